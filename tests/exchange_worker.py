"""One rank of tests/test_gpu_exchange_world2.py: aggregates its rows on the (shared) GPU, runs the library's exchange over
the communicator the library creates from DFX_RCCL_LIB (tests/native/rccl_stub.cpp), writes what it emits as Arrow IPC.
usage: exchange_worker.py <case> <rank> <world> <tmpdir>"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import pyarrow as pa  # noqa: E402

from datafusion_archive_amd import execution as ex  # noqa: E402
import exchange_cases as xc  # noqa: E402

case, rank, world, tmp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
n_keys, pred, group, aggs, opts = xc.CASES[case]
ex.init(0)
for k, v in opts.items():
    ex.set_option(k, v)
# the harness names the injected failure as "<rank>:<stage>" in the environment of the worker PROCESSES it starts; the
# library itself reads no environment variable for it: the worker turns it into the library's test switch
_fail = os.environ.get("DFX_EXCHANGE_FAIL")
if _fail:
    _r, _stage = _fail.split(":")
    ex.set_option("test.exchange_fail", (int(_r) << 8) | xc.EXCHANGE_FAIL_STAGES.index(_stage))
uid_path = os.path.join(tmp, f"uid_{case}")
if rank == 0:
    uid = ex.Communicator.unique_id()
    with open(uid_path + ".tmp", "wb") as f:
        f.write(uid)
    os.rename(uid_path + ".tmp", uid_path)
else:
    for _ in range(600):
        if os.path.exists(uid_path):
            break
        time.sleep(0.05)
    uid = open(uid_path, "rb").read()
comm = ex.Communicator(uid, world, rank)
rel = ex.DataSourceRelation(xc.SCHEMA, xc.batches_of_rank(case, rank))
if pred is not None:
    rel = ex.FilterRelation(rel, ex.compile_scalar_expr(None, pred, xc.SCHEMA), xc.SCHEMA)
rel = ex.AggregateRelation(None, rel, [ex.compile_scalar_expr(None, g, xc.SCHEMA) for g in group],
                           [ex.compile_expr(None, a, xc.SCHEMA) for a in aggs])
if case in xc.FAILURE_CASES:
    try:
        comm.exchange(rel)
        print(f"rank {rank}: NO ERROR")
    except ex.ExecutionError as e:
        print(f"rank {rank}: error: {e.kind}: {e.message}")
    sys.exit(0)
stats = comm.exchange(rel)
out = rel.next()
assert rel.next() is None
with pa.OSFile(os.path.join(tmp, f"out_{case}_{rank}.arrow"), "wb") as sink:
    with pa.ipc.new_file(sink, out.schema) as w:
        w.write_batch(out)
print(f"rank {rank}: {out.num_rows} groups emitted, exchange stats {stats}, collective rounds {ex.counter_get('xchg_rounds')}, "
      f"host syncs {ex.counter_get('xchg_host_syncs')}")
