D=generic-diffusion-feature_amd
cp $D/libgdf.so /tmp/full.so
for v in full mainonly epionly; do
  if [ $v = full ]; then cp /tmp/full.so $D/libgdf.so; else cp $D/libgdf_$v.so $D/libgdf.so; fi
  echo "######## build: $v"; python3 tools/power_epilogue_bound.py 2>&1 | grep -v amdgpu.ids
done
cp /tmp/full.so $D/libgdf.so
