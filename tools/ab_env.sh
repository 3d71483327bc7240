#!/bin/bash
# same-box A/B of one environment switch on bench.py: tools/ab_env.sh VAR "bench args" [rounds]   (VAR=0 vs VAR=1, alternating)
VAR=$1; ARGS=${2:-}; N=${3:-3}
for r in $(seq 1 $N); do
  for v in 0 1; do
    env $VAR=$v python3 bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-extras $ARGS 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); sh = d.get('kernel_time_share', {}); gn = sum(v for k, v in sh.items() if 'gn_' in k)
        print('$VAR=$v run $r:', d['value'], 'img/s', d['ms_per_step'], 'ms', (d.get('power') or {}).get('watts_avg'), 'W; gn kernels share of step', round(gn, 4))"
  done
done
