"""PCIe-inclusive rate of the headline query over HOST Arrow batches (bench.py's host_streamed_pcie_inclusive leg alone)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pyarrow as pa
from datafusion_archive_amd import execution as ex
from datafusion_archive_amd.logicalplan import *
ex.init(0)
schema = pa.schema([("k", pa.int64()), ("v", pa.float64())])
lit = lambda v: Literal(ScalarValue.Float64(v))
pred = BinaryExpr(BinaryExpr(Column(1), Operator.Gt, lit(204.8)), Operator.And, BinaryExpr(Column(1), Operator.Lt, lit(409.6)))
sum_v = AggregateFunction("SUM", [Column(1)], DataType.Float64)
hb_rows = 1 << 24
rng = np.random.default_rng(7)
hb = [pa.RecordBatch.from_arrays([pa.array(rng.integers(0, 1000000, hb_rows).astype(np.int64)),
                                  pa.array(rng.integers(0, 1 << 20, hb_rows).astype(np.float64) / 1024.0)], names=["k", "v"]) for _ in range(4)]
def host_step():
    rel = ex.FilterRelation(ex.DataSourceRelation(schema, hb), ex.compile_scalar_expr(None, pred, schema), schema)
    return ex.AggregateRelation(None, rel, [ex.compile_scalar_expr(None, Column(0), schema)], [ex.compile_expr(None, sum_v, schema)]).next()
host_step(); ex.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); host_step(); host_step(); ex.synchronize(); dt = time.perf_counter() - t0
    gb = 2 * 4 * hb_rows * 16 / dt * 1e-9
    print(f"host-streamed: {dt*1e3:.1f} ms for 2 x 4 x 2^24 rows = {gb:.1f} GB/s = {gb/63:.3f} of PCIe Gen5 x16 (DFX_HOST_PIN={os.environ.get('DFX_HOST_PIN','0')})")
