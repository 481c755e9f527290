#!/usr/bin/env python3
"""CSV source micro-benchmark: a generated numeric file (block repeated to the requested size), parsed on the device.
usage: csv_bench.py [megabytes] [batch_rows] [str]   (str: a fifth, Utf8 column that the query groups by -- the gather kernel runs)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import pyarrow as pa  # noqa: E402
from datafusion_archive_amd import execution as ex  # noqa: E402
from datafusion_archive_amd.logicalplan import AggregateFunction, Column, DataType  # noqa: E402

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
batch_rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 22
with_str = len(sys.argv) > 3 and sys.argv[3] == "str"
rng = np.random.default_rng(1)
n = 20000
block = "\n".join(f"{int(a)},{float(b)!r},{float(c)!r},{int(d)}" + (f",city of {int(a) % 977}" if with_str else "") for a, b, c, d in
                  zip(rng.integers(0, 10**6, n), rng.random(n) * 100, rng.standard_normal(n), rng.integers(-10**9, 10**9, n))) + "\n"
path = "/tmp/dfx_csv_bench.csv"
reps = max(1, mb * 1000000 // len(block))
with open(path, "w") as fh:
    fh.write("k,lat,lng,w,city\n" if with_str else "k,lat,lng,w\n")
    for _ in range(reps):
        fh.write(block)
size = os.path.getsize(path)
rows = reps * n
schema = pa.schema([("k", pa.int64()), ("lat", pa.float64()), ("lng", pa.float64()), ("w", pa.int64())] + ([("city", pa.string())] if with_str else []))
ex.init(0)
aggs = [AggregateFunction("SUM", [Column(1)], DataType.Float64), AggregateFunction("COUNT", [Column(0)], DataType.UInt64)]
keys = [ex.compile_scalar_expr(None, Column(4), schema)] if with_str else []
for it in range(3):
    ex.profile_reset()
    ex.profile_enable(True)
    t0 = time.perf_counter()
    src = ex.CsvDataSource(path, schema, batch_rows)
    t1 = time.perf_counter()
    out = ex.AggregateRelation(None, src, keys, [ex.compile_expr(None, a, schema) for a in aggs]).next()
    ex.synchronize()
    t2 = time.perf_counter()
    ex.profile_enable(False)
    prof = {p["kernel"]: p for p in ex.profile_snapshot()}
    csv = prof.get("csv", {"total_ms": 0, "launches": 0})
    assert sum(out.column(len(keys) + 1).to_pylist()) == rows
    print(f"{size / 1e6:.0f} MB, {rows} rows: open (read + H2D) {1e3 * (t1 - t0):.1f} ms = {size / (t1 - t0) / 1e9:.2f} GB/s; "
          f"index + parse + aggregate {1e3 * (t2 - t1):.1f} ms = {size / (t2 - t1) / 1e9:.2f} GB/s; "
          f"csv kernels {csv['total_ms']:.2f} ms in {csv['launches']} launches = {size / max(csv['total_ms'], 1e-9) / 1e6:.1f} GB/s of text")
    print("   kernels: " + "  ".join(f"{k}:{v['launches']}x{v['total_ms']:.3f}ms" for k, v in prof.items()))
os.remove(path)
