"""The N > 1 path on CPU: world_size 2 over gloo.

`exchange_group_partials` (datafusion_archive_amd/distributed.py) is the code bench.py runs on N
GPUs over RCCL.  Here it runs unchanged over gloo with an oracle-backed stand-in for the three
device entry points (partial_build / partial_export / partial_import), so the bucket layout, the
two all-to-alls, the split sizes and the ownership rule (each group ends on exactly one rank) are
exercised without a GPU.  The stand-in lives in tests/ -- it is a checker, not a product path.
"""
import ctypes
import os
import socket

import numpy as np
import pyarrow as pa
import pytest

import oracle
from datafusion_archive_amd.logicalplan import AggregateFunction, Column, DataType

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

AGGS = [AggregateFunction("sum", [Column(1)], DataType.Float64),
        AggregateFunction("count", [Column(1)], DataType.UInt64),
        AggregateFunction("max", [Column(1)], DataType.Float64)]
MERGE = ["add_f64", "add_u64", "max_f64"]


def _mix(k: np.ndarray) -> np.ndarray:
    z = k.astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


class OracleAgg:
    """Same three entry points as AggregateRelation, computed by the CPU oracle + numpy."""

    def __init__(self, batches):
        self.batches = batches
        self.planes = None  # [n_words][groups] int64 bit patterns
        self.order = None
        self.counts = None

    def partial_build(self, world):
        res = oracle.aggregate([Column(0)], AGGS, self.batches)
        keys = res.column(0).to_numpy().astype(np.int64)
        planes = [keys,
                  res.column(1).to_numpy().astype(np.float64).view(np.int64),
                  res.column(2).to_numpy().astype(np.uint64).view(np.int64),
                  res.column(3).to_numpy().astype(np.float64).view(np.int64)]
        dest = ((_mix(keys) >> np.uint64(7)) % np.uint64(world)).astype(np.int64)
        self.order = np.argsort(dest, kind="stable")
        self.counts = [int((dest == r).sum()) for r in range(world)]
        self.planes = [p[self.order] for p in planes]
        return len(planes), self.counts

    def partial_export(self, dst_ptr, dst_words):
        nw = len(self.planes)
        out = np.empty(nw * sum(self.counts), dtype=np.int64)
        base = 0
        g0 = 0
        for c in self.counts:  # bucket r: word-major planes of its c groups
            for w in range(nw):
                out[base + w * c: base + (w + 1) * c] = self.planes[w][g0:g0 + c]
            base += nw * c
            g0 += c
        assert out.size <= max(dst_words, 0) or out.size == 0
        ctypes.memmove(dst_ptr, out.ctypes.data, out.nbytes)

    def partial_import(self, src_ptr, counts):
        nw = 4
        total = sum(counts)
        buf = np.empty(nw * total, dtype=np.int64)
        ctypes.memmove(buf.ctypes.data, src_ptr, buf.nbytes)
        merged = {}
        base = 0
        for c in counts:
            k = buf[base: base + c]
            s = buf[base + c: base + 2 * c].view(np.float64)
            n = buf[base + 2 * c: base + 3 * c].view(np.uint64)
            m = buf[base + 3 * c: base + 4 * c].view(np.float64)
            for i in range(c):
                key = int(k[i])
                if key in merged:
                    a = merged[key]
                    merged[key] = (a[0] + float(s[i]), a[1] + int(n[i]), max(a[2], float(m[i])))
                else:
                    merged[key] = (float(s[i]), int(n[i]), float(m[i]))
            base += nw * c
        self.merged = merged

    def next(self):
        return self.merged


def _make_data(seed, n):
    rng = np.random.default_rng(seed)
    k = rng.integers(0, 5000, n).astype(np.int64)
    v = rng.integers(0, 2**20, n).astype(np.float64) * 2.0 ** -10
    return pa.RecordBatch.from_arrays([pa.array(k), pa.array(v)], names=["k", "v"])


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from datafusion_archive_amd.distributed import exchange_group_partials
        whole = _make_data(77, 40000)
        per = whole.num_rows // world
        mine = whole.slice(rank * per, per if rank < world - 1 else whole.num_rows - rank * per)
        agg = OracleAgg([mine])
        stats = exchange_group_partials(agg, world, torch.device("cpu"), dist, torch)
        owned = agg.next()
        # every owned key hashes to this rank
        ks = np.array(sorted(owned), dtype=np.int64)
        if ks.size:
            assert np.all(((_mix(ks) >> np.uint64(7)) % np.uint64(world)).astype(np.int64) == rank)
        ret[rank] = (owned, stats)
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_group_partial_exchange_world2_gloo():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    whole = _make_data(77, 40000)
    want = oracle.aggregate([Column(0)], AGGS, [whole])
    want_d = {r[0]: (r[1], r[2], r[3]) for r in zip(*[want.column(i).to_pylist() for i in range(4)])}
    got = {}
    for rank in range(world):
        owned, stats = ret[rank]
        assert not (set(owned) & set(got)), "a group ended up on two ranks"
        got.update(owned)
        assert stats["n_words"] == 4
    assert got == want_d  # exact data: SUM is order-independent, so bit-exact across the exchange
    assert sum(ret[r][1]["sent_groups"] for r in range(world)) == sum(ret[r][1]["received_groups"] for r in range(world))


def _comm_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from datafusion_archive_amd.distributed import library_communicator
        outcome = "created"
        try:
            library_communicator(world, rank, dist)
        except Exception as e:  # no GPU here: the unique id (rank 0) or dfx_comm_init (every rank) fails
            outcome = type(e).__name__ + ": " + str(e)[:160]
        # bench.py's next step: agree on the outcome.  Both ranks must arrive here (nobody is left inside a broadcast)
        t = torch.tensor([1 if outcome == "created" else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ret[rank] = (outcome, int(t.item()))
    finally:
        dist.destroy_process_group()


def test_library_communicator_fails_on_every_rank_together_without_a_gpu():
    """bench.py --gpus N creates the library's RCCL communicator on every rank and then all-reduces whether that worked
    (falling back to the host-driven exchange if not).  Whatever fails -- rank 0's unique id, or dfx_comm_init without a
    device -- every rank has to come out of library_communicator and reach that all-reduce: a rank stuck in the id
    broadcast would hang the job.  World 2 over gloo, no GPU: both ranks report a failure and agree on it."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_comm_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    for rank in range(world):
        outcome, agreed = ret[rank]
        assert agreed == 0
        assert outcome != "created", "a communicator was created without a GPU?"
        assert "Error" in outcome or "error" in outcome, outcome
