#!/usr/bin/env python3
"""PCIe-inclusive rate: the headline query over HOST Arrow batches (uploaded by the library), not a resident table."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import pyarrow as pa  # noqa: E402
from datafusion_archive_amd import execution as ex  # noqa: E402
from datafusion_archive_amd.logicalplan import AggregateFunction, BinaryExpr, Column, DataType, Literal, Operator, ScalarValue  # noqa: E402

rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 27
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 24
ex.init(0)
rng = np.random.default_rng(0)
k = rng.integers(0, 1000000, batch).astype(np.int64)
v = rng.integers(0, 2**20, batch).astype(np.float64) * 2.0 ** -10
b = pa.RecordBatch.from_arrays([pa.array(k), pa.array(v)], names=["k", "v"])
batches = [b] * (rows // batch)
schema = b.schema
lit = lambda x: Literal(ScalarValue.Float64(x))  # noqa: E731
pred = BinaryExpr(BinaryExpr(Column(1), Operator.Gt, lit(204.8)), Operator.And, BinaryExpr(Column(1), Operator.Lt, lit(409.6)))


def run():
    rel = ex.DataSourceRelation(schema, batches)
    rel = ex.FilterRelation(rel, ex.compile_scalar_expr(None, pred, schema), schema)
    rel = ex.AggregateRelation(None, rel, [ex.compile_scalar_expr(None, Column(0), schema)],
                               [ex.compile_expr(None, AggregateFunction("SUM", [Column(1)], DataType.Float64), schema)])
    return rel.next()


run()
for _ in range(3):
    ex.counter_reset()
    t0 = time.perf_counter()
    out = run()
    dt = time.perf_counter() - t0
    moved = ex.counter_get("h2d_bytes")
    print(f"host stream: {rows} rows in {len(batches)} batches of {batch}: {dt*1e3:.1f} ms = {rows/dt/1e9:.2f} G rows/s, "
          f"H2D {moved/1e9:.2f} GB = {moved/dt/1e9:.1f} GB/s, groups={out.num_rows}")
