"""Kernel-time probe of FilterRelation as written (BASELINE config 2): lat = 49 + 10 u, WHERE lat > 51 AND lat < 53,
compacted batches left on the device.  usage: filter_probe.py [rows] [sel=<fraction kept, default 0.2>] [option=value ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyarrow as pa
from datafusion_archive_amd import execution as ex
from datafusion_archive_amd.logicalplan import *
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 30
batch = 1 << 27
sel_want = 0.2
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    if k == "batch": batch = int(v)
    elif k == "sel": sel_want = float(v)
    else: ex.set_option(k, int(v))
ex.init(0)
schema = pa.schema([("lat", pa.float64())])
t = ex.DeviceTable.synth([("lat", ex.SYNTH_F64_UNIFORM, 0, 49.0, 10.0)], 0xDF01, 0, rows)
lit = lambda v: Literal(ScalarValue.Float64(v))
pred = BinaryExpr(BinaryExpr(Column(0), Operator.Gt, lit(51.0)), Operator.And, BinaryExpr(Column(0), Operator.Lt, lit(51.0 + 10.0 * sel_want)))
def run():
    rel = ex.FilterRelation(t.scan(batch), ex.compile_scalar_expr(None, pred, schema), schema)
    return ex.drain_on_device(rel)[0]
kept = run(); ex.synchronize()
t0 = time.perf_counter()
for _ in range(5): run()
ex.synchronize()
dt = (time.perf_counter() - t0) / 5
ex.profile_reset(); ex.profile_enable(True)
for _ in range(3): run()
ex.profile_enable(False)
sel = kept / rows
print(f"filter as written: rows={rows} batch={batch} opts={sys.argv[2:]} kept={kept} ({sel:.4f}): {dt*1e3:.3f} ms per pass = {rows/dt/1e9:.1f} G rows/s = "
      f"{rows*(8.125+8*sel)/dt/1e12:.2f} TB/s algorithmic ({rows*(8.125+8*sel)/dt/8e12:.3f} of 8 TB/s)")
print("   " + "  ".join(f"{p['kernel']}:{p['launches'] // 3}x{p['total_ms'] / p['launches'] * 1e3:.1f}us" for p in ex.profile_snapshot()))
