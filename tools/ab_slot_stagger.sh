# Slot-stagger experiment: libgdf_slot.so = csrc/gemm.hip + tools/experiments/slot_stagger.patch compiled with -DGDF_STAGGER, linked with the other objects
# (tools/build_variant.sh new); copied over libgdf.so for the run.  profiles/r06_ab_deferred_epilogue.txt section 3(c).
D=generic-diffusion-feature_amd
cp $D/libgdf.so /tmp/keep.so; cp $D/libgdf_slot.so $D/libgdf.so
for us in 0 4 8 12 20 30 0; do echo "######## GDF_STAGGER_US=$us (slot stagger, 128-row tiles, 2 workgroups per CU)"; GDF_STAGGER_US=$us python3 tools/bench_slot_stagger.py 2>&1 | grep -v amdgpu.ids; done
cp /tmp/keep.so $D/libgdf.so
