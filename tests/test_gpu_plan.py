"""Scan plans on the device (csrc/dfx_device.hpp: DevScanPlan; PlanPolicy in csrc/dfx_kernels_inl.hpp): the run-time query
shapes -- any single MIN / MAX / COUNT / SUM, one to four `column <op> literal` terms over Int32 ... Float64 columns,
4-byte keys, validity bitmaps on any referenced column -- evaluated as data by the same kernels, against the CPU oracle
(reference-shaped, 1024-row batches) group by group and bit for bit.  Sizes are chosen so that the automatic strategy
takes the partitioned GROUP BY (>= 16384 groups after the calibration slice: pass 1 is the wave-specialised kernel for the
selective scans, the ring kernel for the dense ones); the other strategies' plan kernels run under agg.strategy = 1 / 2.
Every query also runs with scan.plan = 0 (round 3's dispatch: run-time decoded shapes / interpreter / a materialised
FilterRelation for batches with nulls): the two product paths must agree with the oracle and with each other."""
import numpy as np
import pyarrow as pa
import pytest

import oracle
from datafusion_archive_amd import execution as ex
from datafusion_archive_amd.logicalplan import AggregateFunction, BinaryExpr, Column, DataType, Literal, Operator, ScalarValue
from gpu_util import assert_groups_identical, gpu_aggregate
from test_gpu_scale import _assert_bit_exact

pytestmark = pytest.mark.gpu

F64, I64, U64 = DataType.Float64, DataType.Int64, DataType.UInt64
N = (1 << 22) + 12345     # > 2^21 rows: the calibration slice decides the strategy; not a multiple of 64
BATCH = 1 << 21
GROUPS = 200000.0         # >= 16384 groups in the first 2^18 rows: partitioned
SEED = 0xDF04


def f64(v):
    return Literal(ScalarValue.Float64(v))


def i64(v):
    return Literal(ScalarValue.Int64(v))


def i32(v):
    return Literal(ScalarValue.Int32(v))


def AND(*terms):
    e = terms[0]
    for t in terms[1:]:
        e = BinaryExpr(e, Operator.And, t)
    return e


@pytest.fixture(autouse=True)
def _defaults():
    for k, v in (("agg.strategy", 0), ("scan.fast", 1), ("scan.plan", 1), ("agg.capacity_log2", 0), ("agg.partition_mode", 2), ("agg.pass1_ws", 8)):
        ex.set_option(k, v)
    yield
    for k, v in (("agg.strategy", 0), ("scan.fast", 1), ("scan.plan", 1), ("agg.pass1_ws", 8)):
        ex.set_option(k, v)


# columns: k Int64 (or Int32), v Float64 exact (m * 2^-10, m < 2^20), w Int64 in [0, 1000), x Float64 uniform in [0, 1)
def _syn(key_kind=None, v_nulls=0, k_nulls=0, w_nulls=0):
    key_kind = ex.SYNTH_I64_UNIFORM if key_kind is None else key_kind
    return [("k", ex.synth_nulls(key_kind, k_nulls), 0, GROUPS, 0.0), ("v", ex.synth_nulls(ex.SYNTH_F64_EXACT, v_nulls), 1, 0.0, 0.0),
            ("w", ex.synth_nulls(ex.SYNTH_I64_UNIFORM, w_nulls), 2, 1000.0, 0.0), ("x", ex.SYNTH_F64_UNIFORM, 3, 0.0, 1.0)]


def _schema(syn):
    t = {ex.SYNTH_I64_UNIFORM: pa.int64(), ex.SYNTH_I64_ZIPF: pa.int64(), ex.SYNTH_I32_UNIFORM: pa.int32()}
    return pa.schema([(c[0], t.get(c[1] & 0xFF, pa.float64())) for c in syn])


HEAD = AND(BinaryExpr(Column(1), Operator.Gt, f64(204.8)), BinaryExpr(Column(1), Operator.Lt, f64(409.6)))
SUM_V = AggregateFunction("SUM", [Column(1)], F64)
MIN_V = AggregateFunction("MIN", [Column(1)], F64)
MAX_V = AggregateFunction("MAX", [Column(1)], F64)
COUNT_V = AggregateFunction("COUNT", [Column(1)], U64)
SUM_W = AggregateFunction("SUM", [Column(2)], I64)
MAX_W = AggregateFunction("MAX", [Column(2)], I64)


def _run(syn, pred, aggs, plan, opts=()):
    ex.set_option("scan.plan", plan)
    for k, v in opts:
        ex.set_option(k, v)
    t = ex.DeviceTable.synth(syn, SEED, 0, N)
    try:
        return gpu_aggregate([Column(0)], aggs, _schema(syn), [], filter_expr=pred, source=t.scan(BATCH))
    finally:
        ex.set_option("scan.plan", 1)


def _check(name, syn, pred, aggs, opts=()):
    _s, _kept, want = oracle.run_synth_query(syn, SEED, 0, N, 1024, pred, [Column(0)], aggs)
    for plan in (1, 0):
        got = _run(syn, pred, aggs, plan, opts)
        _assert_bit_exact(got, want, f"{name} (scan.plan = {plan})")


CASES = {
    # any single aggregate under the headline's predicate (round 3: SUM only had a fast pass 1)
    "min_only": (_syn(), HEAD, [MIN_V]),
    "max_only": (_syn(), HEAD, [MAX_V]),
    "count_only": (_syn(), HEAD, [COUNT_V]),
    "sum_int64": (_syn(), HEAD, [SUM_W]),
    "max_int64_no_predicate": (_syn(), None, [MAX_W]),
    # predicates: an Int64 column, a third term on the key, four terms over three columns, Eq / NotEq, literal on the left
    "int64_predicate": (_syn(), AND(BinaryExpr(Column(2), Operator.GtEq, i64(200)), BinaryExpr(Column(2), Operator.Lt, i64(400))), [SUM_V]),
    "three_terms": (_syn(), AND(HEAD, BinaryExpr(Column(0), Operator.GtEq, i64(0))), [SUM_V]),
    "four_terms_three_columns": (_syn(), AND(HEAD, BinaryExpr(Column(2), Operator.NotEq, i64(7)), BinaryExpr(f64(0.9), Operator.Gt, Column(3))), [SUM_V]),
    "eq_term": (_syn(), BinaryExpr(Column(2), Operator.Eq, i64(123)), [SUM_V]),
    "one_sided": (_syn(), BinaryExpr(Column(1), Operator.Lt, f64(204.8)), [MIN_V]),
    "impossible_term": (_syn(), AND(HEAD, BinaryExpr(Column(1), Operator.Lt, f64(float("-inf")))), [SUM_V]),
    # 4-byte key
    "int32_key": (_syn(ex.SYNTH_I32_UNIFORM), HEAD, [SUM_V]),
    "int32_key_no_predicate": (_syn(ex.SYNTH_I32_UNIFORM), None, [SUM_V]),
    "int32_key_predicate_on_key": (_syn(ex.SYNTH_I32_UNIFORM), AND(HEAD, BinaryExpr(Column(0), Operator.Lt, i32(150000))), [MAX_V]),
    # validity bitmaps: on the argument (10 %), on the predicate's other column, on the key (value(row) is read blindly)
    "nullable_v_10pct": (_syn(v_nulls=100), HEAD, [SUM_V]),
    "nullable_v_count": (_syn(v_nulls=100), HEAD, [COUNT_V]),
    "nullable_v_lt_keeps_nulls": (_syn(v_nulls=100), BinaryExpr(Column(1), Operator.Lt, f64(204.8)), [COUNT_V]),   # null < x is true (arrow 0.12)
    "nullable_w_predicate": (_syn(w_nulls=300), AND(HEAD, BinaryExpr(Column(2), Operator.NotEq, i64(5))), [SUM_V]),
    "nullable_v_no_filter_count": (_syn(v_nulls=100), None, [COUNT_V]),                                            # no Filter below: COUNT skips nulls
    "nullable_key": (_syn(k_nulls=50), HEAD, [SUM_V]),
    "nullable_everything_int32_key": (_syn(ex.SYNTH_I32_UNIFORM, v_nulls=100, k_nulls=20, w_nulls=500),
                                      AND(HEAD, BinaryExpr(Column(2), Operator.LtEq, i64(900))), [MIN_V]),
}


@pytest.mark.parametrize("name", list(CASES))
def test_plan_family_through_the_partitioned_strategy(name):
    syn, pred, aggs = CASES[name]
    _check(name, syn, pred, aggs)


@pytest.mark.parametrize("name", ["min_only", "int32_key", "nullable_v_10pct", "three_terms"])
def test_plan_family_ring_kernel_and_wide_rows(name):
    """the same queries through the symmetric ring kernel (agg.pass1_ws = 0) and through 16-byte routed rows (agg.narrow_keys = 0)"""
    syn, pred, aggs = CASES[name]
    _check(name + " ring", syn, pred, aggs, opts=(("agg.pass1_ws", 0),))
    ex.set_option("agg.pass1_ws", 8)
    try:
        _check(name + " wide rows", syn, pred, aggs, opts=(("agg.narrow_keys", 0),))
    finally:
        ex.set_option("agg.narrow_keys", -1)


@pytest.mark.parametrize("name", ["min_only", "int32_key", "int32_key_no_predicate", "nullable_v_10pct", "three_terms", "max_int64_no_predicate", "nullable_key"])
def test_plan_family_four_scanners_twelve_routers(name):
    """Round 5: the wave-specialised kernel's DENSE split -- four scanner waves feeding three routers each in turn (one
    single-producer / single-consumer queue per router), chosen for scans that route more than half of their rows; forced here
    (agg.pass1_ws = 4) on selective and unfiltered queries alike, plan kernels and 4-byte keys included."""
    syn, pred, aggs = CASES[name]
    _check(name + " 4 + 12 waves", syn, pred, aggs, opts=(("agg.pass1_ws", 4),))
    ex.set_option("agg.pass1_ws", 8)


def test_signature_queries_four_scanners_twelve_routers():
    """... and the compile-time signatures (the headline with each of its four comparison forms, config 3 without a predicate) through
    the same split, clustered data included (a locally dense stretch after a selective start)"""
    syn = _syn()
    lo, hi = f64(204.8), f64(409.6)
    forms = [AND(BinaryExpr(Column(1), a, lo), BinaryExpr(Column(1), b, hi)) for a in (Operator.Gt, Operator.GtEq) for b in (Operator.Lt, Operator.LtEq)]
    for pred in forms + [None]:
        _s, _kept, want = oracle.run_synth_query(syn, SEED, 0, N, 1024, pred, [Column(0)], [SUM_V])
        ex.set_option("agg.pass1_ws", 4)
        try:
            got = _run(syn, pred, [SUM_V], 1)
        finally:
            ex.set_option("agg.pass1_ws", 8)
        _assert_bit_exact(got, want, f"signature query, 4 + 12 waves, predicate {pred is not None}")


@pytest.mark.parametrize("strategy", [1, 2])
@pytest.mark.parametrize("name", ["nullable_v_count", "nullable_w_predicate", "int32_key", "nullable_everything_int32_key", "nullable_v_no_filter_count"])
def test_plan_family_table_strategies(name, strategy):
    """the global table alone / with the LDS front cache: PlanPolicy inside k_hash_agg (validity bitmaps, 4-byte columns)"""
    syn, pred, aggs = CASES[name]
    _check(f"{name} strategy {strategy}", syn, pred, aggs, opts=(("agg.strategy", strategy),))


def test_plan_two_aggregates_of_different_operands_and_several_keys():
    """several accumulators / two keys (no one-value kernels): the plan looks its slots up (PlanPolicyN)"""
    syn = _syn(v_nulls=100, w_nulls=100)
    pred = AND(HEAD, BinaryExpr(Column(2), Operator.Lt, i64(900)))
    _check("sum_v_max_w", syn, pred, [SUM_V, MAX_W, COUNT_V])  # many groups: a scan per aggregate (agg.split_aggregates, the default)
    _check("sum_v_max_w, one scan", syn, pred, [SUM_V, MAX_W, COUNT_V], opts=(("agg.split_aggregates", 0),))
    ex.set_option("agg.split_aggregates", 1)
    _check("sum_v_max_w, forced partitioned", syn, pred, [SUM_V, MAX_W, COUNT_V], opts=(("agg.strategy", 3),))
    ex.set_option("agg.strategy", 0)
    _check("sum_v_max_w, no predicate", _syn(), None, [SUM_V, MAX_W])
    # two keys: (k mod-free) k and w
    schema = _schema(syn)
    _s, _kept, want = oracle.run_synth_query(syn, SEED, 0, 1 << 20, 1024, pred, [Column(2)], [SUM_V, COUNT_V])
    t = ex.DeviceTable.synth(syn, SEED, 0, 1 << 20)
    got = gpu_aggregate([Column(2)], [SUM_V, COUNT_V], schema, [], filter_expr=pred, source=t.scan(1 << 19))
    _assert_bit_exact(got, want, "few groups (w), nulls")


def test_plan_ungrouped_aggregates_with_nulls_and_int32_columns():
    """k_reduce with a plan: under a Filter every surviving slot is valid (COUNT counts them, SUM adds value(row)); without
    one array_ops::{min,max,sum} skip nulls (aggregate.rs:344-546)"""
    syn = _syn(ex.SYNTH_I32_UNIFORM, v_nulls=100, w_nulls=100)
    schema = _schema(syn)
    for pred in (HEAD, None, AND(HEAD, BinaryExpr(Column(0), Operator.Lt, i32(100000)))):
        aggs = [SUM_V, COUNT_V, MIN_V, MAX_W]
        _s, _kept, want = oracle.run_synth_query(syn, SEED, 0, N, 1024, pred, [], aggs)
        for plan in (1, 0):
            ex.set_option("scan.plan", plan)
            t = ex.DeviceTable.synth(syn, SEED, 0, N)
            got = gpu_aggregate([], aggs, schema, [], filter_expr=pred, source=t.scan(BATCH))
            ex.set_option("scan.plan", 1)
            for i in range(got.num_columns):
                g, w = got.column(i).to_pylist(), want.column(i).to_pylist()
                assert g == w, f"ungrouped, pred {pred is not None}, plan {plan}, aggregate {i}: {g} != {w}"


def test_plan_instead_of_the_compile_time_signatures():
    """scan.plan = 2: the headline and config 3 through the plan kernels instead of their signatures (the A/B bench.py times)"""
    syn = _syn()
    for pred in (HEAD, None):
        _s, _kept, want = oracle.run_synth_query(syn, SEED, 0, N, 1024, pred, [Column(0)], [SUM_V])
        got = _run(syn, pred, [SUM_V], 2)
        _assert_bit_exact(got, want, f"signature query through the plan, predicate {pred is not None}")


def test_wave_specialised_pass1_on_clustered_data_does_not_stall():
    """a scanner published its queue tail once per trip of four row groups, so a locally DENSE stretch after a
    selective calibration slice -- a sorted or time-ordered column under a range predicate -- could fill its queue with rows its
    router could not see yet ('LDS ring stalled').  First 2^18 + 37 rows fail, every later row passes."""
    n = 1 << 22
    rng = np.random.default_rng(7)
    k = rng.integers(0, 200000, n).astype(np.int64)
    v = np.full(n, 300.0)
    v[: (1 << 18) + 37] = 1.0
    v[(1 << 21): (1 << 21) + 5000] = 1.0          # ... and a sparse stretch in the middle, then dense again
    schema = pa.schema([("k", pa.int64()), ("v", pa.float64())])
    b = pa.RecordBatch.from_arrays([pa.array(k), pa.array(v)], schema=schema)
    t = ex.DeviceTable.from_batches(schema, [b])
    for plan in (1, 2):
        ex.set_option("scan.plan", plan)
        got = gpu_aggregate([Column(0)], [SUM_V, COUNT_V], schema, [], filter_expr=HEAD, source=t.scan(1 << 21))
        ex.set_option("scan.plan", 1)
        want = oracle.aggregate([Column(0)], [SUM_V, COUNT_V], [oracle.filter_next(HEAD, b)])
        _assert_bit_exact(got, want, f"clustered data, scan.plan = {plan}")
    # one aggregate (the one-value kernels: this IS the wave-specialised flavour), both splits of its waves
    want = oracle.aggregate([Column(0)], [SUM_V], [oracle.filter_next(HEAD, b)])
    for ws in (8, 4):
        for plan in (1, 2):
            ex.set_option("scan.plan", plan)
            ex.set_option("agg.pass1_ws", ws)
            try:
                got = gpu_aggregate([Column(0)], [SUM_V], schema, [], filter_expr=HEAD, source=t.scan(1 << 21))
            finally:
                ex.set_option("scan.plan", 1)
                ex.set_option("agg.pass1_ws", 8)
            _assert_bit_exact(got, want, f"clustered data, one aggregate, {ws} scanner waves, scan.plan = {plan}")


def test_plan_int32_key_host_batches_at_odd_offsets():
    """Int32 keys from host batches sliced at odd rows (a 4-byte column whose first element is the second half of an aligned
    pair), through the partitioned strategy: the 4-byte-key flavour of the plan kernels (and the widening one, when the
    aggregate is over the key itself)."""
    rng = np.random.default_rng(44)
    n = 600001
    k = rng.integers(-40000, 40000, n).astype(np.int32)
    v = rng.integers(0, 1 << 20, n).astype(np.float64) / 1024.0
    w = rng.integers(0, 1000, n).astype(np.int64)
    wn = rng.random(n) < 0.05
    whole = pa.RecordBatch.from_arrays([pa.array(k), pa.array(v), pa.array(w, mask=wn)], names=["k", "v", "w"])
    batches = [whole.slice(1, 250000), whole.slice(250001, 3), whole.slice(250004, 349997)]
    for name, pred, aggs in (("sum v", HEAD, [SUM_V]), ("min v, w term", AND(HEAD, BinaryExpr(Column(2), Operator.Lt, i64(900))), [MIN_V]),
                             ("max of the key", BinaryExpr(Column(0), Operator.GtEq, i32(-100)), [AggregateFunction("MAX", [Column(0)], DataType.Int32)]),
                             ("no predicate", None, [MAX_W])):
        filtered = [oracle.filter_next(pred, b) for b in batches] if pred is not None else batches
        want = oracle.aggregate([Column(0)], aggs, filtered)
        for strategy in (3, 1):
            ex.set_option("agg.strategy", strategy)
            got = gpu_aggregate([Column(0)], aggs, whole.schema, batches, filter_expr=pred)
            assert_groups_identical(got, want, 1, f"int32 key, odd offsets: {name}, strategy {strategy}")
    ex.set_option("agg.strategy", 0)


def test_per_aggregate_scans_under_skew_growth_and_late_wide_keys():
    """One scan per aggregate (agg.split_aggregates) shares the spill list, the routing scratch and the key plane between the
    aggregates' scans: Zipf keys (regions overflow into the spill list, hot-key pairs), a table that starts at 2^14 slots and
    grows by rehash + replay while two scans feed it, and host batches whose later rows carry keys without a 32-bit image
    (the stream leaves narrow mode between two aggregates' scans).  Against the oracle and against the one-scan form."""
    # (1) Zipf keys, resident table, automatic strategy and a forced small table
    syn = [("k", ex.SYNTH_I64_ZIPF, 0, 1000000.0, 0.0), ("v", ex.SYNTH_F64_EXACT, 1, 0.0, 0.0), ("w", ex.SYNTH_I64_UNIFORM, 2, 1000.0, 0.0)]
    schema = pa.schema([("k", pa.int64()), ("v", pa.float64()), ("w", pa.int64())])
    n = (1 << 22) + 777
    aggs = [SUM_V, MAX_W, COUNT_V]
    _s, _kept, want = oracle.run_synth_query(syn, SEED, 0, n, 1024, HEAD, [Column(0)], aggs)
    for opts in ((), (("agg.capacity_log2", 14),), (("agg.strategy", 3), ("agg.capacity_log2", 14)), (("agg.split_aggregates", 0),)):
        for k, v in opts:
            ex.set_option(k, v)
        try:
            t = ex.DeviceTable.synth(syn, SEED, 0, n)
            got = gpu_aggregate([Column(0)], aggs, schema, [], filter_expr=HEAD, source=t.scan(1 << 20))
            _assert_bit_exact(got, want, f"zipf keys, options {opts}")
        finally:
            for k, _v in opts:
                ex.set_option(k, {"agg.capacity_log2": 0, "agg.strategy": 0, "agg.split_aggregates": 1}[k])
    # (2) wide and negative keys from the third host batch on
    rng = np.random.default_rng(4)
    m = 3 * 700001
    k = rng.integers(0, 300000, m).astype(np.int64)
    k[2 * 700001 + 5::1013] += 1 << 40
    k[2 * 700001 + 9::2027] = -k[2 * 700001 + 9::2027] - 1
    v = rng.integers(0, 1 << 20, m).astype(np.float64) / 1024.0
    w = rng.integers(0, 1000, m).astype(np.int64)
    whole = pa.RecordBatch.from_arrays([pa.array(k), pa.array(v), pa.array(w)], names=["k", "v", "w"])
    batches = [whole.slice(i * 700001, 700001) for i in range(3)]
    want = oracle.aggregate([Column(0)], aggs, [oracle.filter_next(HEAD, b) for b in batches])
    for strategy in (0, 3):
        ex.set_option("agg.strategy", strategy)
        got = gpu_aggregate([Column(0)], aggs, whole.schema, batches, filter_expr=HEAD)
        assert_groups_identical(got, want, 1, f"late wide keys, strategy {strategy}")
    ex.set_option("agg.strategy", 0)
