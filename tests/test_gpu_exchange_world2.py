"""The library's N-rank exchange (dfx_aggregate_exchange, csrc/dfx_exchange.cpp) executed with world = 2 and 3 on ONE GPU:
every rank is a process of its own that loads the product library; DFX_RCCL_LIB points the library's run-time RCCL
binding at tests/native/rccl_stub.cpp, which carries ncclSend / ncclRecv / ncclAllGather between the processes through
host-staged files.  Everything else is the product path: count kernel, grouped sends / receives of counts and buckets,
merge kernels, emit.  The union of what the ranks emit is compared with the CPU oracle over all ranks' rows.

Covers what the single-rank tests cannot: real peers in all_to_all_words, accumulators in several chunks (more than 8),
Utf8 GROUP BY keys (rank-local dictionary ids -> a global dictionary), ungrouped aggregates across ranks.
"""
import os
import subprocess
import sys

import numpy as np
import pyarrow as pa
import pytest

import oracle
import exchange_cases as xc
from gpu_util import assert_groups_identical, assert_batches_identical

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
STUB_SRC = os.path.join(HERE, "native", "rccl_stub.cpp")
STUB_SO = os.path.join(HERE, "native", "librccl_stub.so")


def _build_stub():
    if not os.path.exists(STUB_SO) or os.path.getmtime(STUB_SO) < os.path.getmtime(STUB_SRC):
        subprocess.check_call(["hipcc", "-shared", "-fPIC", "-O1", "-o", STUB_SO, STUB_SRC])
    return STUB_SO


def _run_ranks(case, world, tmp_path, read_results=True, extra_env=None):
    env = dict(os.environ, DFX_RCCL_LIB=_build_stub(), DFX_NO_TORCH="1")
    env.update(extra_env or {})
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "exchange_worker.py"), case, str(r), str(world), str(tmp_path)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for r, p in enumerate(procs):
        try:
            text, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError(f"{case}: rank {r} timed out")
        assert p.returncode == 0, f"{case}: rank {r} failed:\n{text[-3000:]}"
        outs.append(text)
    res = []
    if not read_results:
        return res, outs
    for r in range(world):
        with pa.OSFile(os.path.join(str(tmp_path), f"out_{case}_{r}.arrow"), "rb") as f:
            res.append(pa.ipc.open_file(f).read_all().combine_chunks().to_batches())
    return [b[0] if b else None for b in res], outs


def test_a_failing_rank_ends_the_exchange_on_every_rank(tmp_path):
    """Rank 1 hits DivideByZero while it drains its input.  It still takes part in the first all-to-all and sends the
    failure mark instead of its counts: rank 0 returns an error as well instead of waiting for rank 1's buckets."""
    _res, logs = _run_ranks("peer_failure", 2, tmp_path, read_results=False)
    assert "rank 1: error: ArrowError: DivideByZero" in logs[1], logs[1]
    assert "rank 0: error: ExecutionError" in logs[0] and "rank 1 failed before the exchange" in logs[0], logs[0]


def test_a_failing_rank_ends_the_ungrouped_exchange_on_every_rank(tmp_path):
    """The ungrouped form: rank 1's drain fails (DivideByZero).  Round 3 returned before the all-gather and rank 0 waited for
    ever; now every rank first tells every rank how it is (agree(), one word per peer over the communicator's reserved slab)."""
    _res, logs = _run_ranks("ungrouped_peer_failure", 2, tmp_path, read_results=False)
    assert "rank 1: error: ArrowError: DivideByZero" in logs[1], logs[1]
    assert "rank 0: error: ExecutionError" in logs[0] and "rank 1 failed before the exchange" in logs[0], logs[0]


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", list(xc.FAILURE_STAGES))
def test_a_local_failure_at_any_stage_ends_the_exchange_on_every_rank(case, world, tmp_path):
    """Rank 1 fails locally at a LATER stage (injected: the count kernel, the payload buffers, the last chunk's export, a
    dictionary's blob buffers, the ungrouped merge).  It keeps taking part in every collective with well-formed messages and
    all ranks leave together at the next agree(): nobody hangs (the run would time out), the failing rank reports its own
    error, the others name it."""
    if world == 3 and case not in ("alloc_failure_before_counts", "dict_blob_alloc_failure"):
        pytest.skip("three ranks: one integer-key and one Utf8-key stage")
    _base, stage = xc.FAILURE_STAGES[case]
    _res, logs = _run_ranks(case, world, tmp_path, read_results=False, extra_env={"DFX_EXCHANGE_FAIL": f"1:{stage}"})
    assert f"rank 1: error: ExecutionError" in logs[1] and f"injected failure at stage '{stage}'" in logs[1], logs[1]
    for r in [x for x in range(world) if x != 1]:
        assert f"rank {r}: error: ExecutionError" in logs[r] and "rank 1 failed" in logs[r], logs[r]


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", [c for c in xc.CASES if c not in xc.FAILURE_CASES])
def test_library_exchange_between_processes(case, world, tmp_path):
    if world == 3 and case not in ("int_keys_4_aggs", "utf8_key", "int_keys_lopsided"):
        pytest.skip("three ranks: one integer-key and one Utf8-key case")
    _n_keys, pred, group, aggs, _opts = xc.CASES[case]
    got, logs = _run_ranks(case, world, tmp_path)
    batches = [b for r in range(world) for b in xc.batches_of_rank(case, r)]
    if pred is not None:
        batches = [oracle.filter_next(pred, b) for b in batches]
    want = oracle.aggregate(group, aggs, batches)
    if not group:  # every rank emits the global row
        for r in range(world):
            assert_batches_identical(got[r], want, f"{case} world={world} rank {r}")
        return
    # grouped: every group is emitted by exactly one rank
    present = [b for b in got if b is not None and b.num_rows]
    union = pa.Table.from_batches(present).combine_chunks().to_batches()[0]
    assert sum(b.num_rows for b in present) == want.num_rows, f"{case}: {[b.num_rows for b in present]} groups emitted, oracle has {want.num_rows}\n" + "\n".join(logs)
    assert_groups_identical(union, want, len(group), f"{case} world={world}")
    assert all(b.num_rows > 0 for b in present) and len(present) == world, "every rank owns some groups"
    # round 6: the grouped exchange is TWO collective rounds (the all-gather of states + counts, the buckets) and two host
    # synchronisations; a round more per further chunk of accumulators (more than 8); Utf8 keys add their dictionary rounds
    if case in ("int_keys_4_aggs", "int_keys_partitioned"):
        for r in range(world):
            assert "collective rounds 2, host syncs 2" in logs[r], logs[r]
    if case == "int_keys_lopsided":  # + the agreement after the second allocation (one round, one read-back)
        for r in range(world):
            assert "collective rounds 3, host syncs 3" in logs[r], logs[r]
