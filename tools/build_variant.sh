#!/bin/bash
# Build a diagnostics variant of libgdf.so next to the product library (same-box A/Bs with tools/ab_lib.sh):
#   tools/build_variant.sh <tag> [-DFLAG ...]   ->  generic-diffusion-feature_amd/libgdf_<tag>.so   (git-ignored, travels with gpurun)
# e.g.  tools/build_variant.sh prev -DGDF_CONV_TAP_MAJOR     (round-2 conv K order)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); D=$R/generic-diffusion-feature_amd; TAG=$1; shift
O=$D/build/var_$TAG; mkdir -p $O
pids=()
for f in gemm.hip attn.hip norm.hip dit.hip post.hip model.cpp flux.cpp vae.cpp pixart.cpp api.cpp ops_api.cpp; do
  X=""; [ $f = dit.hip ] && X="-fno-slp-vectorize"       # (as __graft_entry__.PER_FILE_FLAGS)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $X "$@" -x hip -c $D/csrc/$f -o $O/$f.o & pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libgdf_$TAG.so $O/*.o
echo built $D/libgdf_$TAG.so
