"""How the headline step time evolves from a cold process: blocks of 20 steps, timed back to back (tools/gpu_r4_warm_probe.sh).
usage: warm_probe.py [blocks] [option=value ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyarrow as pa
from datafusion_archive_amd import execution as ex
from datafusion_archive_amd.logicalplan import *
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 15
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    ex.set_option(k, int(v))
ex.init(0)
f64 = lambda v: Literal(ScalarValue.Float64(v))
pred = BinaryExpr(BinaryExpr(Column(1), Operator.Gt, f64(204.8)), Operator.And, BinaryExpr(Column(1), Operator.Lt, f64(409.6)))
schema = pa.schema([("k", pa.int64()), ("v", pa.float64())])
t = ex.DeviceTable.synth([("k", ex.SYNTH_I64_UNIFORM, 0, 1e6, 0.0), ("v", ex.SYNTH_F64_EXACT, 1, 0.0, 0.0)], 0xDF00, 0, 1000000000)
def step():
    rel = ex.FilterRelation(t.scan(1 << 27), ex.compile_scalar_expr(None, pred, schema), schema)
    agg = ex.AggregateRelation(None, rel, [ex.compile_scalar_expr(None, Column(0), schema)], [ex.compile_expr(None, AggregateFunction("SUM", [Column(1)], DataType.Float64), schema)])
    return agg.next()
ex.synchronize()
t0 = time.perf_counter(); step(); ex.synchronize(); print(f"cold step {1e3 * (time.perf_counter() - t0):.2f} ms")
for _ in range(5): step()
ex.synchronize()
out = []
for b in range(blocks):
    t0 = time.perf_counter()
    for _ in range(20): step()
    ex.synchronize()
    out.append(1e3 * (time.perf_counter() - t0) / 20)
print("ms per step, blocks of 20 after 5 warm-up steps:", " ".join(f"{x:.3f}" for x in out))
