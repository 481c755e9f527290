# round 5: agg.early_keys with the copy on the DMA engine -- bench headline (no extras) with and without, twice, same box
cd $GRAFT_REPO_ROOT
for r in 1 2; do for o in 1 0; do
  DFX_BENCH_OPTIONS=agg.early_keys=$o timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('early_keys=$o', 'ms_per_step', round(d['ms_per_step'],4), 'e2e', r.get('end_to_end_frac'), 'kernel', r['frac'], 'avg launch', r['avg_launch_ms'])"
done; done
