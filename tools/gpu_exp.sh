#!/bin/bash
for wl in q1 cfg3 cfg2 headline; do python tools/prof_query.py $wl 268435456 3 2>&1 | tail -1; done
python tools/prof_query.py headline 268435456 3 agg.strategy=1 2>&1 | tail -1; python tools/prof_query.py headline 268435456 3 agg.strategy=2 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "grouped" 2>&1 | tail -3
