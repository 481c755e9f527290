"""Host-streamed query (4 x 2^24-row host Arrow batches, the bench's PCIe-inclusive leg) under the host.stream options:
GB/s of H2D per configuration.  usage: host_stream_matrix.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pyarrow as pa
from datafusion_archive_amd import execution as ex
from datafusion_archive_amd.logicalplan import *
ex.init(0)
schema = pa.schema([("k", pa.int64()), ("v", pa.float64())])
rows = 1 << 24
rng = np.random.default_rng(7)
hb = [pa.RecordBatch.from_arrays([pa.array(rng.integers(0, 1000000, rows).astype(np.int64)), pa.array(rng.integers(0, 1 << 20, rows).astype(np.float64) / 1024.0)], names=["k", "v"]) for _ in range(4)]
lit = lambda v: Literal(ScalarValue.Float64(v))
pred = BinaryExpr(BinaryExpr(Column(1), Operator.Gt, lit(204.8)), Operator.And, BinaryExpr(Column(1), Operator.Lt, lit(409.6)))
sum_v = AggregateFunction("SUM", [Column(1)], DataType.Float64)
def step():
    rel = ex.FilterRelation(ex.DataSourceRelation(schema, hb), ex.compile_scalar_expr(None, pred, schema), schema)
    return ex.AggregateRelation(None, rel, [ex.compile_scalar_expr(None, Column(0), schema)], [ex.compile_expr(None, sum_v, schema)]).next()
def measure(tag, n=4):
    step(); ex.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    ex.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{tag:46s} {dt * 1e3:7.2f} ms per 1.07 GB = {4 * rows * 16 / dt * 1e-9:5.1f} GB/s ({4 * rows * 16 / dt * 1e-9 / 63.0:.3f} of the link)", flush=True)
print("host cores:", os.cpu_count())
ex.set_option("host.stream", 0); measure("mode 0: in order, pageable")
ex.set_option("host.stream", 1)
for threads in (2, 4, 8, 12, 16):
    for piece, slots in ((8, 8), (2, 16), (16, 6), (32, 4)):
        ex.set_option("host.stage_threads", threads); ex.set_option("host.stage_mb", piece); ex.set_option("host.stage_slots", slots)
        measure(f"mode 1: {threads} threads, {piece} MB x {slots} slots")
ex.set_option("host.stream", 2); measure("mode 2: one batch ahead")
