for i in $(seq 1 16); do
  GDF_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600+i)) bench.py --gpus 2 --steps 2 --warmup 1 --version 1-5 --batch 4 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('run', $i, r['kernel'], 'launches', r['launches'], 'achieved', r['achieved'], 'avg_ms', r.get('avg_launch_ms'), 'graph', d.get('hipgraph', {}).get('launches_in_timed_region'), d.get('hipgraph', {}).get('eager_fallbacks_in_timed_region'))"
done
