#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sort.py -q --timeout 600 2>&1 | tail -n 12
python - <<'PY'
import sys, time
sys.path.insert(0, ".")
import pyarrow as pa
from datafusion_archive_amd import execution as ex
from datafusion_archive_amd.logicalplan import Column
ex.init(0)
n = 1 << 26
syn = [("k", ex.SYNTH_I64_UNIFORM, 0, float(2**62), 0.0), ("v", ex.SYNTH_F64_EXACT, 1, 0.0, 0.0)]
schema = pa.schema([("k", pa.int64()), ("v", pa.float64())])
t = ex.DeviceTable.synth(syn, 1, 0, n)
for it in range(2):
    ex.profile_reset(); ex.profile_enable(True)
    t0 = time.perf_counter()
    rel = ex.SortRelation(t.scan(1 << 24), [(ex.compile_scalar_expr(None, Column(0), schema), True)], schema)
    lim = ex.LimitRelation(rel, 10, schema)
    out = lim.next(); ex.synchronize()
    dt = time.perf_counter() - t0
    ex.profile_enable(False)
    p = {x["kernel"]: x for x in ex.profile_snapshot()}
    print(f"sort 2^26 rows (i64 key, 8 digit passes) + f64 payload: {dt*1e3:.1f} ms = {n/dt/1e9:.2f} G rows/s; sort kernels {p['sort']['total_ms']:.1f} ms in {p['sort']['launches']} launches")
PY
