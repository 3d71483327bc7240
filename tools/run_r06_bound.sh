D=generic-diffusion-feature_amd
python -m pytest tests/test_gpu_diffusers_branch.py -q -s 2>&1 | tail -150 > gpurun_out/r06_diffusers_branch.log
cp $D/libgdf.so /tmp/full.so
for v in full mainonly epionly full; do
  if [ $v = full ]; then cp /tmp/full.so $D/libgdf.so; else cp $D/libgdf_$v.so $D/libgdf.so; fi
  echo "######## build: $v"; python3 tools/bench_epilogue_bound.py
done > gpurun_out/r06_epilogue_bound_shapes.txt 2>&1
cp /tmp/full.so $D/libgdf.so
tail -8 gpurun_out/r06_diffusers_branch.log
