#!/bin/bash
# BOUND the prize of any "deferred / overlapped epilogue" scheme before building one.
# Three builds of libgdf.so on the same box (tools/build_variant.sh, run HERE first):
#   full      the product library
#   mainonly  -DGDF_ABLATE_EPI=1: every GEMM / conv kernel WITHOUT its epilogue (accumulators kept alive, nothing loaded or stored)
#   epionly   -DGDF_ABLATE_EPI=2: every GEMM / conv kernel WITHOUT its main loop (epilogue traffic on zero accumulators)
# Per op class: T_full, T_main, T_epi from the synchronising per-op profile.  A perfect overlap leaves max(T_main, T_epi); the prize is
# T_full - max(T_main, T_epi).  (Results of the ablated builds are garbage by construction; only times are read.)
#   tools/build_variant.sh mainonly -DGDF_ABLATE_EPI=1; tools/build_variant.sh epionly -DGDF_ABLATE_EPI=2
#   gpurun -- 'bash tools/ab_epilogue_bound.sh > gpurun_out/r06_epilogue_bound.txt 2>&1'
D=generic-diffusion-feature_amd
cp $D/libgdf.so /tmp/full.so
for v in full mainonly epionly full; do
  if [ $v = full ]; then cp /tmp/full.so $D/libgdf.so; else cp $D/libgdf_$v.so $D/libgdf.so; fi
  echo "######## build: $v"
  echo "==== VAE encode / decode, 16 x 1024^2 (per-op profile of one sub-batch)"
  python3 tools/bench_vae.py --steps 3 2>&1 | grep -v "^$"
  echo "==== SD1.5 512^2 B=32 per-op table"
  python3 tools/op_table.py 1-5 32 2>&1 | head -40
  echo "==== SDXL 1024^2 B=16 per-op table"
  python3 tools/op_table.py xl 16 2>&1 | head -40
done
cp /tmp/full.so $D/libgdf.so
