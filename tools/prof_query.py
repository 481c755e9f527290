#!/usr/bin/env python3
"""Runs one workload a few times (no CPU baseline): the target of rocprofv3 runs.
usage: prof_query.py <headline|selNN|cfg3|cfg2|q1|neighbour|oneterm|threeterm|threecol|diffop|avgmax|product> [rows] [iters] [option=value ...]
(neighbour: SELECT k, SUM(v), MIN(v) WHERE v >= lo AND v < hi GROUP BY k -- two aggregates of one operand;
 oneterm: SELECT k, SUM(v) WHERE v < 204.8 GROUP BY k; threecol: SELECT k, SUM(w) WHERE v > lo AND v < hi GROUP BY k;
 product: SELECT k, SUM(v * 2.0) WHERE v > lo AND v < hi GROUP BY k -- shapes without a static signature: FastPolicy)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyarrow as pa  # noqa: E402
from datafusion_archive_amd import execution as ex  # noqa: E402
from datafusion_archive_amd.logicalplan import (AggregateFunction, BinaryExpr, Column, DataType, Literal, Operator,  # noqa: E402
                                                ScalarValue)

wl = sys.argv[1] if len(sys.argv) > 1 else "headline"
rows = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1 << 28
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
batch_rows = 1 << 26
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    if k == "batch":
        batch_rows = int(v)
    else:
        ex.set_option(k, int(v))
ex.init(0)
f64 = DataType.Float64


def lit(v):
    return Literal(ScalarValue.Float64(v))


if wl == "q1":
    syn = [("rf", ex.SYNTH_I64_UNIFORM, 0, 3.0, 0.0), ("ls", ex.SYNTH_I64_UNIFORM, 1, 2.0, 0.0),
           ("qty", ex.SYNTH_F64_UNIFORM, 2, 1.0, 49.0), ("price", ex.SYNTH_F64_UNIFORM, 3, 900.0, 104100.0),
           ("disc", ex.SYNTH_F64_UNIFORM, 4, 0.0, 0.10), ("tax", ex.SYNTH_F64_UNIFORM, 5, 0.0, 0.08),
           ("ship", ex.SYNTH_F64_UNIFORM, 6, 0.0, 2526.0)]
    schema = pa.schema([(n, pa.int64() if i < 2 else pa.float64()) for i, (n, *_r) in enumerate(syn)])
    one_minus = BinaryExpr(lit(1.0), Operator.Minus, Column(4))
    one_plus = BinaryExpr(lit(1.0), Operator.Plus, Column(5))
    dp = BinaryExpr(Column(3), Operator.Multiply, one_minus)
    aggs = [AggregateFunction("sum", [Column(2)], f64), AggregateFunction("sum", [Column(3)], f64),
            AggregateFunction("sum", [dp], f64), AggregateFunction("sum", [BinaryExpr(dp, Operator.Multiply, one_plus)], f64)]
    pred = BinaryExpr(BinaryExpr(Column(6), Operator.LtEq, lit(2436.0)), Operator.And,
                      BinaryExpr(Column(4), Operator.GtEq, lit(0.0)))
    group = [Column(0), Column(1)]
    bytes_per_row = 56
else:
    syn = [("k", ex.SYNTH_I64_UNIFORM, 0, 1e6, 0.0), ("v", ex.SYNTH_F64_EXACT, 1, 0.0, 0.0)]
    if os.environ.get("ZIPF") == "1":  # Zipf(1.0)-distributed keys (bench.py's zipf_keys leg)
        syn[0] = ("k", ex.SYNTH_I64_ZIPF, 0, 1e6, 1.0)
    schema = pa.schema([("k", pa.int64()), ("v", pa.float64())])
    pred = BinaryExpr(BinaryExpr(Column(1), Operator.Gt, lit(204.8)), Operator.And,
                      BinaryExpr(Column(1), Operator.Lt, lit(409.6)))
    aggs = [AggregateFunction("SUM", [Column(1)], f64)]
    group = [Column(0)]
    bytes_per_row = 16
    if wl == "neighbour":
        pred = BinaryExpr(BinaryExpr(Column(1), Operator.GtEq, lit(204.8)), Operator.And,
                          BinaryExpr(Column(1), Operator.Lt, lit(409.6)))
        aggs = [AggregateFunction("SUM", [Column(1)], f64), AggregateFunction("MIN", [Column(1)], f64)]
    if wl == "oneterm":
        pred = BinaryExpr(Column(1), Operator.Lt, lit(204.8))
    if wl == "threeterm":  # a third term on the Int64 key: no compile-time signature (FastPolicy)
        pred = BinaryExpr(pred, Operator.And, BinaryExpr(Column(0), Operator.GtEq, Literal(ScalarValue.Int64(0))))
    if wl == "threecol":
        syn = syn + [("w", ex.SYNTH_F64_EXACT, 2, 0.0, 0.0)]
        schema = pa.schema([("k", pa.int64()), ("v", pa.float64()), ("w", pa.float64())])
        aggs = [AggregateFunction("SUM", [Column(2)], f64)]
        bytes_per_row = 24
    if wl == "diffop":  # two aggregates of DIFFERENT operands: generic 24-byte routed rows
        syn = syn + [("w", ex.SYNTH_F64_EXACT, 2, 0.0, 0.0)]
        schema = pa.schema([("k", pa.int64()), ("v", pa.float64()), ("w", pa.float64())])
        aggs = [AggregateFunction("SUM", [Column(1)], f64), AggregateFunction("MIN", [Column(2)], f64)]
        bytes_per_row = 24
    if wl == "avgmax":  # three accumulators over two columns: AVG(v) = SUM + COUNT, MAX(w)
        syn = syn + [("w", ex.SYNTH_F64_EXACT, 2, 0.0, 0.0)]
        schema = pa.schema([("k", pa.int64()), ("v", pa.float64()), ("w", pa.float64())])
        aggs = [AggregateFunction("AVG", [Column(1)], f64), AggregateFunction("MAX", [Column(2)], f64)]
        bytes_per_row = 24
    if wl == "product":
        aggs = [AggregateFunction("SUM", [BinaryExpr(Column(1), Operator.Multiply, lit(2.0))], f64)]
    if wl.startswith("sel"):  # sel50 / sel80 / sel35 ...: the headline query keeping that share of the rows (v is uniform on [0, 1024))
        pred = BinaryExpr(BinaryExpr(Column(1), Operator.Gt, lit(204.8)), Operator.And,
                          BinaryExpr(Column(1), Operator.Lt, lit(204.8 + 10.24 * float(wl[3:]))))
    if wl == "cfg3" or os.environ.get("NOPRED") == "1":  # (NOPRED=1: any of the workloads above without its predicate: every row routed)
        pred = None
    if wl == "cfg2":
        group = []
        aggs = [AggregateFunction("COUNT", [Column(1)], DataType.UInt64)]
        bytes_per_row = 8

table = ex.DeviceTable.synth(syn, 0xDF02, 0, rows)


def run():
    rel = table.scan(batch_rows)
    if pred is not None:
        rel = ex.FilterRelation(rel, ex.compile_scalar_expr(None, pred, schema), schema)
    rel = ex.AggregateRelation(None, rel, [ex.compile_scalar_expr(None, g, schema) for g in group],
                               [ex.compile_expr(None, a, schema) for a in aggs])
    return rel.next()


run()
ex.synchronize()
ex.counter_reset()
ex.profile_reset()
ex.profile_enable(os.environ.get('NOPROF') != '1')
t0 = time.perf_counter()
for _ in range(iters):
    out = run()
ex.synchronize()
dt = (time.perf_counter() - t0) / iters
ex.profile_enable(False)
prof = " ".join(f"{p['kernel']}:{p['launches'] // iters}x{p['total_ms'] / iters:.3f}ms" for p in ex.profile_snapshot()
                if p["total_ms"] / iters > 0.02)
print("   host us/iter: " + " ".join(f"{c[4:-3]}:{ex.counter_get(c) / iters:.0f}" for c in ("agg_drain_us", "agg_emit_us", "agg_ctrl_wait_us", "agg_sync_us", "agg_alloc_us")))
print(f"   kernels/iter: {prof}")
print(f"{wl}: rows={rows} {dt*1e3:.3f} ms/iter  {rows/dt/1e9:.2f} Grows/s  {rows*bytes_per_row/dt/1e9:.1f} GB/s  groups={out.num_rows}")
