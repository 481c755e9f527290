#!/usr/bin/env python3
"""Two (or more) processes run the headline query over their own resident tables on ONE GPU at the same time: is the sum of
their rates above one process's?  (Would pipelining windows over several streams pay: pass 1 of one window next to pass 2
of another, no launch gaps, no tails.)  usage: lanes_probe.py <rows> <iters> <rank> <world> <dir> [workload]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyarrow as pa
from datafusion_archive_amd import execution as ex
from datafusion_archive_amd.logicalplan import *
rows = int(float(sys.argv[1])); iters = int(sys.argv[2]); rank = int(sys.argv[3]); world = int(sys.argv[4]); d = sys.argv[5]
wl = sys.argv[6] if len(sys.argv) > 6 else "headline"
ex.init(0)
syn = [("k", ex.SYNTH_I64_UNIFORM, 0, 1e6, 0.0), ("v", ex.SYNTH_F64_EXACT, 1, 0.0, 0.0)]
schema = pa.schema([("k", pa.int64()), ("v", pa.float64())])
t = ex.DeviceTable.synth(syn, 0xDF02 + rank, 0, rows)
lit = lambda v: Literal(ScalarValue.Float64(v))
pred = BinaryExpr(BinaryExpr(Column(1), Operator.Gt, lit(204.8)), Operator.And, BinaryExpr(Column(1), Operator.Lt, lit(409.6)))
def run():
    rel = t.scan(1 << 27)
    if wl == "headline":
        rel = ex.FilterRelation(rel, ex.compile_scalar_expr(None, pred, schema), schema)
    rel = ex.AggregateRelation(None, rel, [ex.compile_scalar_expr(None, Column(0), schema)], [ex.compile_expr(None, AggregateFunction("SUM", [Column(1)], DataType.Float64), schema)])
    return rel.next()
run(); run(); ex.synchronize()
open(os.path.join(d, f"ready{rank}"), "w").close()
while not all(os.path.exists(os.path.join(d, f"ready{r}")) for r in range(world)):
    time.sleep(0.0005)
t0 = time.perf_counter()
per = []
for _ in range(iters):
    a = time.perf_counter(); run(); ex.synchronize(); per.append((time.perf_counter() - a) * 1e3)
dt = time.perf_counter() - t0
s = sorted(per)
print(f"{wl} rank {rank}/{world}: rows={rows} iters={iters} total {dt*1e3:.1f} ms, median {s[len(s)//2]:.2f} ms/iter, start {t0:.4f} end {t0+dt:.4f}  {rows*iters/dt/1e9:.1f} Grows/s", flush=True)
