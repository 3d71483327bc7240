#!/bin/bash
# (see also tools/ab_slot_stagger.sh: the per-CU slot form for the two-workgroups-per-CU tiles)
# De-phasing experiment (csrc/kernels.h GemmParams::stagger): same library (built with -DGDF_STAGGER: tools/build_variant.sh stagger -DGDF_STAGGER,
# then copied over libgdf.so on the GPU box), same box, the shape-level bench under several start delays.
for cfg in "0 2" "3 2" "6 2" "12 2" "4 4" "8 4" "3 8" "6 8" "0 2"; do
  set -- $cfg
  echo "######## GDF_STAGGER_US=$1 GDF_STAGGER_GROUPS=$2"
  GDF_STAGGER_US=$1 GDF_STAGGER_GROUPS=$2 python3 tools/bench_epilogue_bound.py 2>&1 | grep -v amdgpu.ids
done
