#!/bin/bash
# GroupNorm finalize + apply in ONE launch (csrc/norm.hip gn_finalize_apply_kernel) against the separate gn_finalize + gn_apply
# launches, same library, same box, alternating.  GDF_GN_FINALIZE_APPLY=0 -> A (two launches), =1 -> B (one launch).
for r in 1 2 3; do
  for v in 0 1; do
    echo "== GDF_GN_FINALIZE_APPLY=$v run $r"
    GDF_GN_FINALIZE_APPLY=$v python3 bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  sdxl B=16   ', d['value'], 'img/s', d['ms_per_step'], 'ms', {k: v for k, v in d['kernel_time_share'].items() if 'gn_' in k})"
    GDF_GN_FINALIZE_APPLY=$v python3 bench.py --version 1-5 --batch 32 --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  sd1.5 B=32  ', d['value'], 'img/s', d['ms_per_step'], 'ms')"
    GDF_GN_FINALIZE_APPLY=$v python3 tools/bench_vae.py --steps 5 2>/dev/null | grep "^VAE"
  done
done
