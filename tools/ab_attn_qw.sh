#!/bin/bash
# same-box A/B of the one-wave-per-SIMD attention experiment (csrc/attn.hip, GDF_ATTN_QW4): product library vs the variants built by
#   tools/build_variant.sh qw3 -DGDF_ATTN_QW4=3 ; tools/build_variant.sh qw4 -DGDF_ATTN_QW4=4 ; tools/build_variant.sh w8 -DGDF_ATTN_QW4=28
#   VARIANTS="prod w8" bash tools/ab_attn_qw.sh
D=generic-diffusion-feature_amd
cp $D/libgdf.so /tmp/prod.so
for r in 1 2; do
  for v in ${VARIANTS:-prod qw3 qw4}; do
    if [ $v = prod ]; then cp /tmp/prod.so $D/libgdf.so; else cp $D/libgdf_$v.so $D/libgdf.so; fi
    echo "== $v run $r"
    python tools/bench_attn_quick.py 2>/dev/null | grep sdxl
    python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print('bench', d['value'], 'img/s', d['ms_per_step'], 'ms', (d.get('power') or {}).get('watts_avg'), 'W')"
  done
done
for v in ${VARIANTS:-prod qw3 qw4}; do
  [ $v = prod ] && continue
  cp $D/libgdf_$v.so $D/libgdf.so; echo "== $v parity"; python -m pytest tests/test_gpu_ops.py -q -m gpu -k "attn or attention" 2>&1 | tail -2
done
cp /tmp/prod.so $D/libgdf.so
