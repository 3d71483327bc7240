#!/bin/bash
# Diagnostics: attention kernel with parts compiled out (csrc/attn.hip GDF_ATTN_ABLATE bits: 1 no K/V loads + LDS stores in the
# loop, 2 no softmax arithmetic, 4 (with 1) no workgroup barrier).  Build HERE, then on the GPU box: python tools/ablate_attn.py
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/generic-diffusion-feature_amd/csrc; B=$R/generic-diffusion-feature_amd/build; O=$R/tools/micro/build
mkdir -p $O
for a in 1 2 3 5 7 8 16 128; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DGDF_ATTN_ABLATE=$a -x hip -c $C/attn.hip -o $O/attn_abl$a.o &
done
for pr in 0 2; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DGDF_ATTN_PRIO=$pr -x hip -c $C/attn.hip -o $O/attn_prio$pr.o &
done
wait
for pr in 0 2; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libgdf_attn_prio$pr.so $O/attn_prio$pr.o $(ls $B/*.o | grep -v attn.hip.o)
done
for a in 1 2 3 5 7 8 16 128; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libgdf_attn_abl$a.so $O/attn_abl$a.o $(ls $B/*.o | grep -v attn.hip.o)
done
for tr in 0 32 64 96; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DGDF_ATTN_TRACE -DGDF_ATTN_ABLATE=$tr -x hip -c $C/attn.hip -o $O/attn_trace$tr.o &
done
wait
for tr in 0 32 64 96; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libgdf_attn_trace$tr.so $O/attn_trace$tr.o $(ls $B/*.o | grep -v attn.hip.o)
done
cp $O/libgdf_attn_trace0.so $O/libgdf_attn_trace.so
ls $O/libgdf_attn_abl*.so $O/libgdf_attn_trace.so
