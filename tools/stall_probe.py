#!/usr/bin/env python3
"""Per-step times of the headline query over a resident table, with and without the key column copied ahead of time
(agg.early_keys: a DMA-engine copy on the side stream): does a step ever stall, and on which setting?
usage: stall_probe.py [rows] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyarrow as pa
from datafusion_archive_amd import execution as ex
from datafusion_archive_amd.logicalplan import *
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**10
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
ex.init(0)
syn = [("k", ex.SYNTH_I64_UNIFORM, 0, 1e6, 0.0), ("v", ex.SYNTH_F64_EXACT, 1, 0.0, 0.0)]
schema = pa.schema([("k", pa.int64()), ("v", pa.float64())])
t = ex.DeviceTable.synth(syn, 0xDF02, 0, rows)
lit = lambda v: Literal(ScalarValue.Float64(v))
pred = BinaryExpr(BinaryExpr(Column(1), Operator.Gt, lit(204.8)), Operator.And, BinaryExpr(Column(1), Operator.Lt, lit(409.6)))
def run():
    rel = ex.FilterRelation(t.scan(1 << 27), ex.compile_scalar_expr(None, pred, schema), schema)
    rel = ex.AggregateRelation(None, rel, [ex.compile_scalar_expr(None, Column(0), schema)], [ex.compile_expr(None, AggregateFunction("SUM", [Column(1)], DataType.Float64), schema)])
    return rel.next()
for early in ((0, 1, 0, 1) if os.environ.get('STALL_ORDER') == '0101' else (1, 0, 1, 0)):
    ex.set_option("agg.early_keys", early)
    run(); ex.synchronize()
    per = []
    for _ in range(steps):
        t0 = time.perf_counter(); run(); ex.synchronize(); per.append((time.perf_counter() - t0) * 1e3)
    s = sorted(per)
    print(f"early_keys={early} rows={rows}: median {s[len(s)//2]:.2f} ms, min {s[0]:.2f}, max {s[-1]:.2f}; steps over 1.15 x median: "
          + " ".join(f"#{i}:{x:.1f}" for i, x in enumerate(per) if x > 1.15 * s[len(s)//2]) + "  first five: " + " ".join(f"{x:.1f}" for x in per[:5])
          + f"  late copies {ex.counter_get('agg_early_keys_late')}", flush=True)
