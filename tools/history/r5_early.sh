# round 5: the key column downloaded while the scan runs (agg.early_keys) -- its test, and the headline / config 3 with and
# without it, alternating in ONE process order (two rounds) so that box drift does not pass for an effect
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "key_column_downloaded" 2>&1 | tail -8
export KPROBE_BATCH_LOG2=27 KPROBE_QUERIES=8
for r in 1 2; do for o in 1 0; do timeout 300 python tools/kprobe.py 1e9 1e6 1 agg.early_keys=$o agg.merge_scan_batches=1 2>&1 | grep "per-query" | cut -c1-120; done; done
for r in 1 2; do for o in 1 0; do timeout 300 python tools/kprobe.py 1e9 1e6 0 agg.early_keys=$o agg.merge_scan_batches=1 2>&1 | grep "per-query\|partition:" | cut -c1-200; done; done
