"""Cases of the multi-rank exchange tests (tests/test_gpu_exchange_world2.py), shared by the parent and the rank processes.

Every case is (schema, batches_of_rank(rank, world) -> [RecordBatch], filter expr or None, group exprs, aggregate exprs).
The rows of rank r are rows [r * ROWS, (r + 1) * ROWS) of one deterministic table, so the parent can run the CPU oracle
over the concatenation and compare it with the union of what the ranks emit (every group is emitted by exactly one rank).
"""
import numpy as np
import pyarrow as pa

from datafusion_archive_amd.logicalplan import AggregateFunction, BinaryExpr, Column, DataType, Literal, Operator, ScalarValue

F64, U64, I64 = DataType.Float64, DataType.UInt64, DataType.Int64
ROWS = 1 << 19


def lit(v):
    return Literal(ScalarValue.Float64(v))


def _table(rank, n_keys, seed=11):
    """k Int64 below n_keys, v = m * 2^-10 (exact sums), w = small integers as f64, s = a string derived from k"""
    rng = np.random.default_rng(seed + rank)
    k = rng.integers(0, n_keys, ROWS).astype(np.int64)
    v = rng.integers(0, 1 << 20, ROWS).astype(np.float64) / 1024.0
    w = rng.integers(-50, 50, ROWS).astype(np.float64)
    s = pa.array([("key-%d" % x) * (1 + x % 3) for x in (k % 5003)])
    return pa.RecordBatch.from_arrays([pa.array(k), pa.array(v), pa.array(w), s], names=["k", "v", "w", "s"])


SCHEMA = pa.schema([("k", pa.int64()), ("v", pa.float64()), ("w", pa.float64()), ("s", pa.string())])
PRED = BinaryExpr(BinaryExpr(Column(1), Operator.Gt, lit(204.8)), Operator.And, BinaryExpr(Column(1), Operator.Lt, lit(819.2)))


def agg(name, e, t=F64):
    return AggregateFunction(name, [e], t)


_Q11 = [agg("sum", Column(1)), agg("sum", Column(2)), agg("sum", BinaryExpr(Column(1), Operator.Multiply, lit(2.0))),
        agg("min", Column(1)), agg("max", Column(1)), agg("count", Column(1), U64), agg("min", Column(2)), agg("max", Column(2)),
        agg("sum", BinaryExpr(Column(2), Operator.Plus, lit(1.0))), agg("count", Column(2), U64), agg("avg", Column(2))]

CASES = {
    # name: (n_keys, filter, group, aggregates, force options)
    "int_keys_4_aggs": (100000, PRED, [Column(0)], [agg("sum", Column(1)), agg("count", Column(1), U64), agg("min", Column(1)), agg("max", Column(2))], {}),
    "int_keys_partitioned": (100000, PRED, [Column(0)], [agg("sum", Column(1))], {"agg.strategy": 3}),
    # ranks above 0 hold seven keys but OWN half (a third ...) of rank 0's hundred thousand: they receive far more groups than the
    # capacity they announced in round 1, so every rank takes the extra allocate-and-agree round (round 6's protocol)
    "int_keys_lopsided": (100000, None, [Column(0)], [agg("sum", Column(1)), agg("max", Column(2))], {}),
    "ungrouped": (1000, PRED, [], [agg("sum", Column(1)), agg("count", Column(1), U64), agg("min", Column(2)), agg("max", Column(1))], {}),
    "eleven_accumulators": (3000, PRED, [Column(0)], _Q11, {}),
    "eleven_accumulators_ungrouped": (3000, None, [], _Q11, {}),
    "utf8_key": (20000, PRED, [Column(3)], [agg("sum", Column(1)), agg("count", Column(1), U64), agg("max", Column(2))], {}),
    "utf8_and_int_keys": (700, None, [Column(3), Column(0)], [agg("sum", Column(2)), agg("min", Column(1))], {}),
}


# a rank-local error (DivideByZero while rank 1 drains its input) must end the exchange on EVERY rank, not leave the others
# waiting for buckets that never come
CASES["peer_failure"] = (1000, None, [Column(0)], [agg("sum", BinaryExpr(Column(1), Operator.Divide, Column(2)))], {})


# ... the same for ungrouped aggregates: round 3 returned before the all-gather, the peers waited for ever
CASES["ungrouped_peer_failure"] = (1000, None, [], [agg("sum", BinaryExpr(Column(1), Operator.Divide, Column(2))), agg("count", Column(1), U64)], {})
# stage numbers of dfx_set_option("test.exchange_fail", rank << 8 | stage) (csrc/dfx_exchange.cpp: kFailStages)
EXCHANGE_FAIL_STAGES = ["", "drain", "count", "payload_alloc", "export", "dict_local", "dict_blob_alloc", "merge"]
# local failures INJECTED at the later stages (the worker's DFX_EXCHANGE_FAIL = "1:<stage>" -> test.exchange_fail): the
# count kernel, the payload buffers, the last export, a dictionary's blob buffers, the ungrouped merge -- the query and the
# data are those of a passing case
FAILURE_STAGES = {"alloc_failure_before_counts": ("int_keys_4_aggs", "count"), "payload_alloc_failure": ("int_keys_4_aggs", "payload_alloc"),
                  "export_failure": ("eleven_accumulators", "export"), "dict_blob_alloc_failure": ("utf8_key", "dict_blob_alloc"),
                  "ungrouped_merge_failure": ("eleven_accumulators_ungrouped", "merge")}
for _name, (_base, _stage) in FAILURE_STAGES.items():
    CASES[_name] = CASES[_base]
FAILURE_CASES = ["peer_failure", "ungrouped_peer_failure"] + list(FAILURE_STAGES)


def batches_of_rank(case, rank):
    n_keys = CASES[case][0]
    b = _table(rank, 7 if (case == "int_keys_lopsided" and rank > 0) else n_keys)
    if case in ("peer_failure", "ungrouped_peer_failure"):
        w = np.ones(ROWS)
        if rank == 1:
            w[7] = 0.0
        b = pa.RecordBatch.from_arrays([b.column(0), b.column(1), pa.array(w), b.column(3)], names=["k", "v", "w", "s"])
    return [b.slice(0, ROWS // 2), b.slice(ROWS // 2)]  # two batches per rank
