"""Kernel-time probe: runs one query a few times and prints the built-in profiler's per-kernel averages."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyarrow as pa
from datafusion_archive_amd import execution as ex
from datafusion_archive_amd.logicalplan import *
rows = int(float(sys.argv[1])); groups = float(sys.argv[2]); filt = int(sys.argv[3])
LO, HI = 204.8, 409.6
UNGROUPED = False
ZIPF = False
WIDE = False
for kv in sys.argv[4:]:
    if kv == "ungrouped": UNGROUPED = True; continue
    if kv == "zipf": ZIPF = True; continue
    if kv == "wide": WIDE = True; continue
    if kv.startswith("lo="): LO = float(kv[3:]); continue
    if kv.startswith("hi="): HI = float(kv[3:]); continue
    k, v = kv.split("="); ex.set_option(k, int(v))
ex.init(0)
syn = [("k", ex.SYNTH_I64_WIDE if WIDE else ex.SYNTH_I64_ZIPF if ZIPF else ex.SYNTH_I64_UNIFORM, 0, groups, 1.0 if ZIPF else 0.0), ("v", ex.SYNTH_F64_EXACT, 1, 0.0, 0.0)]
schema = pa.schema([("k", pa.int64()), ("v", pa.float64())])
t = ex.DeviceTable.synth(syn, 0xDF02, 0, rows)
lit = lambda v: Literal(ScalarValue.Float64(v))
pred = BinaryExpr(BinaryExpr(Column(1), Operator.Gt, lit(LO)), Operator.And, BinaryExpr(Column(1), Operator.Lt, lit(HI)))
def run():
    rel = t.scan(int(os.environ.get("KPROBE_BATCH_LOG2", "26")) and 1 << int(os.environ.get("KPROBE_BATCH_LOG2", "26")))
    if filt: rel = ex.FilterRelation(rel, ex.compile_scalar_expr(None, pred, schema), schema)
    if UNGROUPED:
        rel = ex.AggregateRelation(None, rel, [], [ex.compile_expr(None, AggregateFunction("COUNT", [Column(1)], DataType.UInt64), schema)])
    else:
        rel = ex.AggregateRelation(None, rel, [ex.compile_scalar_expr(None, Column(0), schema)], [ex.compile_expr(None, AggregateFunction("SUM", [Column(1)], DataType.Float64), schema)])
    return rel.next()
import time
run(); ex.synchronize()
per = []
ex.counter_reset()
import gc
if os.environ.get("KPROBE_NOGC") == "1": gc.disable()
NQ = int(os.environ.get("KPROBE_QUERIES", "5"))
marks = []
for _ in range(NQ):
    c0 = {k: ex.counter_get(k) for k in ("agg_drain_us", "agg_emit_us", "export_us", "export_alloc_us")}
    t0 = time.perf_counter(); out = run(); ex.synchronize(); per.append((time.perf_counter() - t0) * 1e3)
    marks.append("/".join(str(ex.counter_get(k) - c0[k]) for k in ("agg_drain_us", "agg_emit_us", "export_us", "export_alloc_us")))
if max(per) > 1.5 * min(per): print("   slow query breakdown (drain/emit/export/export-alloc us per query): " + " ".join(marks))
dt = sum(per) / len(per) / 1e3
print("per-query ms: " + " ".join(f"{x:.2f}" for x in per) + "   host us/query: " + " ".join(
    f"{k}={ex.counter_get('agg_' + k) / len(per):.0f}" for k in ("ctrl_wait_us", "sync_us", "emit_us", "drain_us", "alloc_us", "pass2_launches", "growths")) + f" export_us={ex.counter_get('export_us') / len(per):.0f}")
ex.profile_reset(); ex.profile_enable(True)
for _ in range(3): out = run()
ex.profile_enable(False)
print(f"un-instrumented: {dt * 1e3:.3f} ms per query = {rows / dt / 1e9:.1f} G rows/s")
print(f"rows={rows} groups={groups} filt={filt} opts={sys.argv[4:]} -> groups_out={out.num_rows}")
print("   " + "  ".join(f"{p['kernel']}:{p['launches'] // 3}x{p['total_ms'] / p['launches'] * 1e3:.1f}us" for p in ex.profile_snapshot()))
