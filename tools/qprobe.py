"""Query probe: per-kernel times (the library's HIP-event profiler) and end-to-end time of named query variants over a
synthetic resident table.  usage: qprobe.py <rows> <variant>[,<variant>...] [option=value ...]
variants: headline | plan (headline, scan.plan = 2) | min | count | int64pred | three | int32key | nullv | nullv_count | dense |
          dense_plan | reject (nothing passes) | reject_plan | interp (scan.fast = 0) | sum_min_w (two operands)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyarrow as pa
from datafusion_archive_amd import execution as ex
from datafusion_archive_amd.logicalplan import *

rows = int(float(sys.argv[1]))
variants = sys.argv[2].split(",")
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    ex.set_option(k, int(v))
ex.init(0)
G = 1e6
f64 = lambda v: Literal(ScalarValue.Float64(v))
i64 = lambda v: Literal(ScalarValue.Int64(v))
AND = lambda a, b: BinaryExpr(a, Operator.And, b)
HEAD = AND(BinaryExpr(Column(1), Operator.Gt, f64(204.8)), BinaryExpr(Column(1), Operator.Lt, f64(409.6)))
REJECT = AND(BinaryExpr(Column(1), Operator.Gt, f64(5000.0)), BinaryExpr(Column(1), Operator.Lt, f64(6000.0)))
SUM_V = AggregateFunction("SUM", [Column(1)], DataType.Float64)
KV = [("k", ex.SYNTH_I64_UNIFORM, 0, G, 0.0), ("v", ex.SYNTH_F64_EXACT, 1, 0.0, 0.0)]
K32V = [("k", ex.SYNTH_I32_UNIFORM, 0, G, 0.0), KV[1]]
KNV = [KV[0], ("v", ex.synth_nulls(ex.SYNTH_F64_EXACT, 100), 1, 0.0, 0.0)]
KVW = KV + [("w", ex.SYNTH_F64_EXACT, 2, 0.0, 0.0)]
S = lambda syn: pa.schema([(c[0], pa.int32() if (c[1] & 0xFF) == ex.SYNTH_I32_UNIFORM else pa.int64() if (c[1] & 0xFF) in (ex.SYNTH_I64_UNIFORM, ex.SYNTH_I64_ZIPF) else pa.float64()) for c in syn])
Q = {  # name: (columns, predicate, aggregates, options)
    "headline": (KV, HEAD, [SUM_V], {}), "plan": (KV, HEAD, [SUM_V], {"scan.plan": 2}),
    "min": (KV, HEAD, [AggregateFunction("MIN", [Column(1)], DataType.Float64)], {}),
    "count": (KV, HEAD, [AggregateFunction("COUNT", [Column(1)], DataType.UInt64)], {}),
    "int64pred": (KV, AND(BinaryExpr(Column(0), Operator.GtEq, i64(200000)), BinaryExpr(Column(0), Operator.Lt, i64(400000))), [SUM_V], {}),
    "three": (KV, AND(HEAD, BinaryExpr(Column(0), Operator.GtEq, i64(0))), [SUM_V], {}),
    "int32key": (K32V, HEAD, [SUM_V], {}), "nullv": (KNV, HEAD, [SUM_V], {}),
    "nullv_count": (KNV, HEAD, [AggregateFunction("COUNT", [Column(1)], DataType.UInt64)], {}),
    "dense": (KV, None, [SUM_V], {}), "dense_plan": (KV, None, [SUM_V], {"scan.plan": 2}),
    "reject": (KV, REJECT, [SUM_V], {}), "reject_plan": (KV, REJECT, [SUM_V], {"scan.plan": 2}),
    "interp": (KV, HEAD, [SUM_V], {"scan.fast": 0}),
    "sum_min_w": (KVW, HEAD, [SUM_V, AggregateFunction("MIN", [Column(2)], DataType.Float64)], {}),
}
tables = {}
for name in variants:
    syn, pred, aggs, opts = Q[name]
    key = id(syn)
    if key not in tables:
        tables.clear()  # one table in HBM at a time
        tables[key] = ex.DeviceTable.synth(syn, 0xDF02, 0, rows)
    t, schema = tables[key], S(syn)
    for k, v in opts.items():
        ex.set_option(k, v)

    def run():
        rel = t.scan(1 << 27)
        if pred is not None:
            rel = ex.FilterRelation(rel, ex.compile_scalar_expr(None, pred, schema), schema)
        return ex.AggregateRelation(None, rel, [ex.compile_scalar_expr(None, Column(0), schema)], [ex.compile_expr(None, a, schema) for a in aggs]).next()
    try:
        run(); run(); ex.synchronize()
        n = 5
        t0 = time.perf_counter()
        for _ in range(n):
            out = run()
        ex.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        ex.profile_reset(); ex.profile_enable(True)
        for _ in range(3):
            run()
        ex.profile_enable(False)
        prof = "  ".join(f"{p['kernel']}:{p['launches'] // 3}x{p['total_ms'] / p['launches'] * 1e3:.1f}us" for p in ex.profile_snapshot() if p["kernel"] in ("partition", "partition_agg", "hash_agg"))
        print(f"{name:12s} {ms:8.3f} ms/query  {rows / ms / 1e6:7.1f} G rows/s  groups={out.num_rows}  {prof}", flush=True)
    finally:
        for k in opts:
            ex.set_option(k, 1)
