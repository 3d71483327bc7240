#!/bin/bash
# same-box A/B of two builds on the headline step: generic-diffusion-feature_amd/libgdf_prev.so (A) vs libgdf.so (B), alternating, N rounds (default 3)
D=generic-diffusion-feature_amd; N=${1:-3}
cp $D/libgdf.so /tmp/new.so
for r in $(seq 1 $N); do
  for v in prev new; do
    if [ $v = prev ]; then cp $D/libgdf_prev.so $D/libgdf.so; else cp /tmp/new.so $D/libgdf.so; fi
    python3 bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print('$v run $r:', d['value'], 'img/s', d['ms_per_step'], 'ms', (d.get('power') or {}).get('watts_avg'), 'W', d['roofline']['achieved'], 'TF dominant kernel')"
  done
done
cp /tmp/new.so $D/libgdf.so
