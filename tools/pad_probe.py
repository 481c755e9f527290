#!/usr/bin/env python3
"""Does padding between the producers' region sets (agg.partition_pad) change pass 1, placement held fixed?  One process; per trial
the scratch is re-created behind a dummy allocation, then the SAME buffer serves pad = P (allocated first, the larger layout) and
pad = 0, alternating.  usage: pad_probe.py <rows> <filter 0|1>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyarrow as pa
from datafusion_archive_amd import execution as ex
from datafusion_archive_amd.logicalplan import *
rows = int(float(sys.argv[1])); filt = int(sys.argv[2])
ex.init(0)
ex.set_option("agg.early_keys", 1)
syn = [("k", ex.SYNTH_I64_UNIFORM, 0, 1e6, 0.0), ("v", ex.SYNTH_F64_EXACT, 1, 0.0, 0.0)]
schema = pa.schema([("k", pa.int64()), ("v", pa.float64())])
lit = lambda v: Literal(ScalarValue.Float64(v))
pred = BinaryExpr(BinaryExpr(Column(1), Operator.Gt, lit(204.8)), Operator.And, BinaryExpr(Column(1), Operator.Lt, lit(409.6)))
def query(t):
    rel = t.scan(1 << 27)
    if filt: rel = ex.FilterRelation(rel, ex.compile_scalar_expr(None, pred, schema), schema)
    rel = ex.AggregateRelation(None, rel, [ex.compile_scalar_expr(None, Column(0), schema)],
                               [ex.compile_expr(None, AggregateFunction("SUM", [Column(1)], DataType.Float64), schema)])
    return rel.next()
def measure(t):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); query(t); ex.synchronize(); best = min(best, (time.perf_counter() - t0) * 1e3)
    ex.profile_reset(); ex.profile_enable(True); query(t); ex.profile_enable(False)
    p = {x["kernel"]: x for x in ex.profile_snapshot()}
    return best, p["partition"]["total_ms"] / p["partition"]["launches"] * 1e3, p["partition_agg"]["total_ms"] / p["partition_agg"]["launches"] * 1e3
t = ex.DeviceTable.synth(syn, 0xDF02, 0, rows)
dummy_syn = [("x", ex.SYNTH_I64_UNIFORM, 0, 10.0, 0.0)]
for trial in range(6):
    P = 12480 if trial % 2 == 0 else (1 << 20) + 4288
    ex.set_option("pool.trim", 1)
    dummy = ex.DeviceTable.synth(dummy_syn, 1, 0, (trial * 7 + 3) << 20)
    line = []
    for pad in (P, 0, P, 0):
        ex.set_option("agg.partition_pad", pad)
        query(t); ex.synchronize()
        b, p1, p2 = measure(t)
        line.append(f"pad {pad}: {b:.3f} ms p1 {p1:.1f} p2 {p2:.1f}")
    del dummy
    print(f"placement {trial}: " + " | ".join(line), flush=True)
