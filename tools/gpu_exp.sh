#!/bin/bash
for rep in 1 2; do for b in 67108864 134217728; do python bench.py --no-cpu-baseline --no-extras --batch-rows $b --steps 8 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['extra']['kernels']; print(d['config']['batch_rows'], round(d['value']/1e9,1), round(d['ms_per_step'],3), d['roofline']['frac'], 'kernel_ms_per_step', round(sum(v['total_ms'] for v in k.values())/d['steps'],3))"; done; done
