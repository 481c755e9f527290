#!/usr/bin/env python3
"""Does the step time of the headline depend on WHERE its buffers landed?  One process: the same query after (a) the library's
cached device blocks were dropped and a dummy allocation of a varying size was put in front of the new ones (routing scratch,
GROUP BY table, result buffers move; the resident table stays), (b) the resident table itself was created again behind a dummy.
Prints the best of 5 runs per placement.  profiles/r05_placement_probe.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyarrow as pa
from datafusion_archive_amd import execution as ex
from datafusion_archive_amd.logicalplan import *
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**9
ex.init(0)
for kv in sys.argv[2:]:
    k, v = kv.split("="); ex.set_option(k, int(v))
syn = [("k", ex.SYNTH_I64_UNIFORM, 0, 1e6, 0.0), ("v", ex.SYNTH_F64_EXACT, 1, 0.0, 0.0)]
schema = pa.schema([("k", pa.int64()), ("v", pa.float64())])
lit = lambda v: Literal(ScalarValue.Float64(v))
pred = BinaryExpr(BinaryExpr(Column(1), Operator.Gt, lit(204.8)), Operator.And, BinaryExpr(Column(1), Operator.Lt, lit(409.6)))
def query(t):
    rel = ex.FilterRelation(t.scan(1 << 27), ex.compile_scalar_expr(None, pred, schema), schema)
    rel = ex.AggregateRelation(None, rel, [ex.compile_scalar_expr(None, Column(0), schema)],
                               [ex.compile_expr(None, AggregateFunction("SUM", [Column(1)], DataType.Float64), schema)])
    return rel.next()
def best(t, n=5):
    out = []
    for _ in range(n):
        t0 = time.perf_counter(); query(t); ex.synchronize(); out.append((time.perf_counter() - t0) * 1e3)
    return min(out), out
def pass1(t):
    ex.profile_reset(); ex.profile_enable(True)
    query(t); query(t)
    ex.profile_enable(False)
    p = {x["kernel"]: x for x in ex.profile_snapshot()}
    return p["partition"]["total_ms"] / p["partition"]["launches"] * 1e3, p["partition_agg"]["total_ms"] / p["partition_agg"]["launches"] * 1e3
dummy_syn = [("x", ex.SYNTH_I64_UNIFORM, 0, 10.0, 0.0)]
t = ex.DeviceTable.synth(syn, 0xDF02, 0, rows)
query(t); ex.synchronize()
for trial in range(7):
    b, all_ = best(t)
    p1, p2 = pass1(t)
    print(f"scratch placement {trial}: best {b:.3f} ms  (runs {' '.join(f'{x:.2f}' for x in all_)})  pass 1 {p1:.1f} us  pass 2 {p2:.1f} us", flush=True)
    ex.set_option("pool.trim", 1)
    dummy = ex.DeviceTable.synth(dummy_syn, 1, 0, (trial * 7 + 3) << 20)   # 24 MB + 56 MB per trial, kept alive while the scratch is reallocated
    query(t); ex.synchronize()
    del dummy
for trial in range(5):
    del t
    ex.set_option("pool.trim", 1)
    dummy = ex.DeviceTable.synth(dummy_syn, 1, 0, (trial * 11 + 5) << 20)
    t = ex.DeviceTable.synth(syn, 0xDF02, 0, rows)
    del dummy
    query(t); ex.synchronize()
    b, all_ = best(t)
    p1, p2 = pass1(t)
    print(f"table placement {trial}: best {b:.3f} ms  (runs {' '.join(f'{x:.2f}' for x in all_)})  pass 1 {p1:.1f} us  pass 2 {p2:.1f} us", flush=True)
