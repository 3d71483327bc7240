#!/bin/bash
# Diagnostics: where does the two-group 256x320 main loop spend its time?  Builds libgdf variants with parts of the (four-phase)
# loop compiled out (csrc/gemm.hip, GDF_ABLATE bits: 1 no fragment reads, 2 no LDS-DMA, 4 no barriers, 8 no MFMAs) HERE (no GPU
# needed), then on the GPU box:   python tools/ablate_gemm.py      -> TFLOP/s-equivalent per variant
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/generic-diffusion-feature_amd/csrc; B=$R/generic-diffusion-feature_amd/build; O=$R/tools/micro/build
mkdir -p $O
for a in 1 2 3 4 8 9 10 11 12; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DGDF_ABLATE=$a -x hip -c $C/gemm.hip -o $O/gemm_abl$a.o &
done
# the round-1 four-phase schedules of both two-group kernels, for the bit-exact A/B against the shipped two-phase ones
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DGDF_PHASES4 -x hip -c $C/gemm.hip -o $O/gemm_phases4.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libgdf_phases4.so $O/gemm_phases4.o $(ls $B/*.o | grep -v gemm.hip.o)
for a in 1 2 3 4 8 9 10 11 12; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libgdf_abl$a.so $O/gemm_abl$a.o $(ls $B/*.o | grep -v gemm.hip.o)
done
ls -la $O/*.so
