#!/bin/bash
# Diagnostics: libgdf with workgroup time stamps in the GEMM kernels (csrc/gemm.hip, -DGDF_TRACE) -> tools/micro/build/libgdf_trace.so
# Build HERE (no GPU needed), then on the GPU box:  python tools/trace_gemm.py
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/generic-diffusion-feature_amd/csrc; B=$R/generic-diffusion-feature_amd/build; O=$R/tools/micro/build
mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DGDF_TRACE -x hip -c $C/gemm.hip -o $O/gemm_trace.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libgdf_trace.so $O/gemm_trace.o $(ls $B/*.o | grep -v gemm.hip.o)
ls -la $O/libgdf_trace.so
