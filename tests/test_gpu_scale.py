"""Per-group parity AT THE BENCHMARK'S SIZE: 2^28 rows, 1 M Int64 keys, 2^26-row batches --
the configuration bench.py's headline runs (auto strategy => partitioned: pass 1 routes rows to table blocks, pass 2
aggregates the blocks in LDS), compared with the CPU oracle KEY BY KEY, not with another GPU result.

The oracle (oracle/dfx_oracle.c, reference-shaped: 1024-row batches, row-at-a-time hash map, aggregate.rs:787-952 and
:548-612) needs 15 s (filtered) to ~2 min (every row through the hash map) per query at this size on one host core, so
all the oracle queries of this module are started together on their own host threads when the first test asks for one
(the C code holds no global state and ctypes releases the GIL); each test then waits for its own.
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pyarrow as pa
import pytest

import oracle
from datafusion_archive_amd import execution as ex
from datafusion_archive_amd.logicalplan import AggregateFunction, BinaryExpr, Column, DataType, Literal, Operator, ScalarValue
from gpu_util import gpu_aggregate

pytestmark = pytest.mark.gpu

F64 = DataType.Float64
N = 1 << 28
N_DENSE = 1 << 27  # the unfiltered variants (every row routed: two routing windows of 2^26 rows; the oracle needs 30 s per 2^27 such rows)
BATCH = 1 << 26
SEED = 0xDF02
SCHEMA = pa.schema([("k", pa.int64()), ("v", pa.float64())])
SUM_V = AggregateFunction("SUM", [Column(1)], F64)
COUNT_V = AggregateFunction("COUNT", [Column(1)], DataType.UInt64)


def lit(v):
    return Literal(ScalarValue.Float64(v))


PRED = BinaryExpr(BinaryExpr(Column(1), Operator.Gt, lit(204.8)), Operator.And,
                  BinaryExpr(Column(1), Operator.Lt, lit(409.6)))


@pytest.fixture(autouse=True)
def _defaults():
    for k, v in (("agg.strategy", 0), ("scan.fast", 1), ("agg.capacity_log2", 0), ("agg.partition_mode", 2)):
        ex.set_option(k, v)
    yield


def _sorted_columns(batch: pa.RecordBatch):
    """(keys, [columns...]) sorted by key, as numpy arrays (floats stay floats; compared by bit pattern later)."""
    k = batch.column(0).to_numpy()
    order = np.argsort(k, kind="stable")
    return k[order], [batch.column(i).to_numpy(zero_copy_only=False)[order] for i in range(1, batch.num_columns)]


SYN_UNIFORM_KEYS_EXACT = [("k", ex.SYNTH_I64_UNIFORM, 0, 1e6, 0.0), ("v", ex.SYNTH_F64_EXACT, 1, 0.0, 0.0)]
SYN_UNIFORM_VALUES = [("k", ex.SYNTH_I64_UNIFORM, 0, 1e6, 0.0), ("v", ex.SYNTH_F64_UNIFORM, 1, 0.0, 1.0)]
SYN_ZIPF = [("k", ex.SYNTH_I64_ZIPF, 0, 1e6, 1.0), ("v", ex.SYNTH_F64_EXACT, 1, 0.0, 0.0)]
# 10^6 distinct keys (u + 1) * 0x9E3779B97F4A7C15 mod 2^64: every one >= 2^32 (no 32-bit hash image: 16-byte routed rows, 64-bit
# key compares in pass 2), half of them negative -- the reference takes any Int64 key (aggregate.rs:807-852)
SYN_WIDE = [("k", ex.SYNTH_I64_WIDE, 0, 1e6, 0.0), ("v", ex.SYNTH_F64_EXACT, 1, 0.0, 0.0)]
PRED_U = BinaryExpr(BinaryExpr(Column(1), Operator.Gt, lit(0.2)), Operator.And, BinaryExpr(Column(1), Operator.Lt, lit(0.4)))
MINMAX = [SUM_V, AggregateFunction("MIN", [Column(1)], F64), AggregateFunction("MAX", [Column(1)], F64)]
QUERIES = {
    "headline": (SYN_UNIFORM_KEYS_EXACT, PRED, [SUM_V, COUNT_V]),
    "config3": (SYN_UNIFORM_KEYS_EXACT, None, MINMAX),
    "uniform_filtered": (SYN_UNIFORM_VALUES, PRED_U, [SUM_V, COUNT_V]),
    "uniform_all": (SYN_UNIFORM_VALUES, None, [SUM_V, COUNT_V]),
    "zipf_filtered": (SYN_ZIPF, PRED, [SUM_V, COUNT_V]),
    "zipf_all": (SYN_ZIPF, None, [SUM_V, COUNT_V]),
    "wide_filtered": (SYN_WIDE, PRED, [SUM_V, COUNT_V]),
    "wide_all": (SYN_WIDE, None, [SUM_V]),
}
# ---- BASELINE config 2 as written and config 5 (TPC-H Q1 shape) at the benchmark's batch sizes ----------------------
N2 = 1 << 28                 # config 2: one Float64 column, lat = 49 + 10 u, seed 0xDF01, 2^27-row batches
BATCH2 = 1 << 27
SEED2 = 0xDF01
SYN_LAT = [("lat", ex.SYNTH_F64_UNIFORM, 0, 49.0, 10.0)]
SCHEMA2 = pa.schema([("lat", pa.float64())])
PRED2 = BinaryExpr(BinaryExpr(Column(0), Operator.Gt, lit(51.0)), Operator.And, BinaryExpr(Column(0), Operator.Lt, lit(53.0)))
N5 = 1 << 27                 # config 5: 7 columns (56 B/row), 2 predicates, 2 keys, 4 SUMs of expressions, 6 groups
SEED5 = 0xDF05
# exact-arithmetic variant: qty, price = m * 2^-2 (12 bits), disc, tax in {0, .25, .5, .75}: every product has <= 18
# significant bits and every partial sum over 2^27 rows <= 45 -- any summation order gives the same doubles
SYN_Q1_EXACT = [("rf", ex.SYNTH_I64_UNIFORM, 0, 3.0, 0.0), ("ls", ex.SYNTH_I64_UNIFORM, 1, 2.0, 0.0),
                ("qty", ex.SYNTH_F64_EXACT, 2, 12.0, 2.0), ("price", ex.SYNTH_F64_EXACT, 3, 12.0, 2.0),
                ("disc", ex.SYNTH_F64_EXACT, 4, 2.0, 2.0), ("tax", ex.SYNTH_F64_EXACT, 5, 2.0, 2.0),
                ("ship", ex.SYNTH_F64_UNIFORM, 6, 0.0, 2526.0)]
# bench.py's columns (uniform doubles: sums are order-dependent, checked within the stated tolerance)
SYN_Q1_UNIFORM = SYN_Q1_EXACT[:2] + [("qty", ex.SYNTH_F64_UNIFORM, 2, 1.0, 49.0), ("price", ex.SYNTH_F64_UNIFORM, 3, 900.0, 104100.0),
                                     ("disc", ex.SYNTH_F64_UNIFORM, 4, 0.0, 0.10), ("tax", ex.SYNTH_F64_UNIFORM, 5, 0.0, 0.08),
                                     ("ship", ex.SYNTH_F64_UNIFORM, 6, 0.0, 2526.0)]
SCHEMA5 = pa.schema([(c[0], pa.int64() if i < 2 else pa.float64()) for i, c in enumerate(SYN_Q1_EXACT)])
_DP = BinaryExpr(Column(3), Operator.Multiply, BinaryExpr(lit(1.0), Operator.Minus, Column(4)))
AGGS5 = [AggregateFunction("sum", [Column(2)], F64), AggregateFunction("sum", [Column(3)], F64), AggregateFunction("sum", [_DP], F64),
         AggregateFunction("sum", [BinaryExpr(_DP, Operator.Multiply, BinaryExpr(lit(1.0), Operator.Plus, Column(5)))], F64)]
PRED5 = BinaryExpr(BinaryExpr(Column(6), Operator.LtEq, lit(2436.0)), Operator.And, BinaryExpr(Column(4), Operator.GtEq, lit(0.0)))

_pool = None
_futures = {}


def _rows(name):
    """rows of QUERIES[name]: the filtered queries run at the size bench.py's headline is checked at (2^28), the unfiltered ones at 2^27"""
    return N_DENSE if QUERIES[name][1] is None else N


def _oracle_result(name):
    """(seconds, rows kept, result batch) of the oracle for QUERIES[name]; every query is started on first use."""
    global _pool
    if _pool is None:
        _pool = ThreadPoolExecutor(len(QUERIES) + 3)
        for q, (syn, pred, aggs) in QUERIES.items():
            _futures[q] = _pool.submit(oracle.run_synth_query, syn, SEED, 0, _rows(q), 1024, pred, [Column(0)], aggs)
        # config 2: FilterRelation reference-shaped, 1024-row batches, compacted column + the predicate's BooleanArray
        _futures["cfg2_filter"] = _pool.submit(oracle.run_synth_filter, SYN_LAT, SEED2, 0, N2, 1024, PRED2)
        for q, syn in (("q1_exact", SYN_Q1_EXACT), ("q1_uniform", SYN_Q1_UNIFORM)):  # + COUNT: rows per group for the tolerance
            _futures[q] = _pool.submit(oracle.run_synth_query, syn, SEED5, 0, N5, 1024, PRED5, [Column(0), Column(1)], AGGS5 + [COUNT_V])
    return _futures[name].result()


def _gpu(name):
    syn, pred, aggs = QUERIES[name]
    t = ex.DeviceTable.synth(syn, SEED, 0, _rows(name))
    return gpu_aggregate([Column(0)], aggs, SCHEMA, [], filter_expr=pred, source=t.scan(BATCH))


def _assert_keys_equal(gk, wk, what):
    assert len(gk) == len(wk), f"{what}: {len(gk)} groups != {len(wk)}"
    assert len(np.unique(gk)) == len(gk), f"{what}: duplicate groups on the device"
    assert np.array_equal(gk, wk), f"{what}: key sets differ"


def _assert_bit_exact(got, want, what):
    gk, gc = _sorted_columns(got)
    wk, wc = _sorted_columns(want)
    _assert_keys_equal(gk, wk, what)
    for i, (g, w) in enumerate(zip(gc, wc)):
        gb = g.view(np.uint64) if g.dtype == np.float64 else g
        wb = w.view(np.uint64) if w.dtype == np.float64 else w
        bad = np.nonzero(gb != wb)[0]
        assert bad.size == 0, f"{what}: aggregate {i}: {bad.size} groups differ, e.g. key {gk[bad[0]]}: got {g[bad[0]]!r} want {w[bad[0]]!r}"


def test_headline_query_per_group_vs_oracle_at_2_28_rows():
    """bench.py's query (filter + GROUP BY SUM over the exact distribution) on 2^28 rows: every one of the 10^6 groups
    has the oracle's SUM bit for bit and the oracle's COUNT."""
    got = _gpu("headline")
    _secs, kept, want = _oracle_result("headline")
    assert got.num_rows == 1000000
    assert int(got.column(2).to_numpy().sum()) == kept
    _assert_bit_exact(got, want, "headline 2^28")


def test_config3_no_filter_per_group_vs_oracle_at_2_27_rows():
    """BASELINE config 3 as written (SELECT k, SUM(v) GROUP BY k, no filter: every row is routed), plus MIN/MAX."""
    got = _gpu("config3")
    _secs, kept, want = _oracle_result("config3")
    assert kept == _rows("config3") and got.num_rows == 1000000
    _assert_bit_exact(got, want, "config 3 2^27")


def _exact_sums_uniform(pred_lo_hi, n):
    """correctly rounded EXACT SUM(v) of every group of the first n rows of the `uniform` table (oracle.ExactGroupSums), slice by slice"""
    ex_ = oracle.ExactGroupSums(1000000)
    step = 1 << 24
    for r0 in range(0, n, step):
        k = oracle.synth_column(oracle.SYNTH_I64_UNIFORM, 0, 1e6, 0.0, SEED, r0, min(step, n - r0))
        v = oracle.synth_column(oracle.SYNTH_F64_UNIFORM, 1, 0.0, 1.0, SEED, r0, min(step, n - r0))
        if pred_lo_hi is not None:
            keep = (v > pred_lo_hi[0]) & (v < pred_lo_hi[1])
            k, v = k[keep], v[keep]
        ex_.add(k, v)
    return ex_.result()


def test_uniform_values_within_tolerance_at_2_28_rows():
    """The `uniform` variant (v in [0, 1): partial sums are NOT exact, so a parallel sum cannot reproduce the
    reference's sequential rounding).  Per group with n rows (tests/oracle.py: check_float_sums):
    |gpu - reference| <= n * eps * sum|v| (proven: the same n terms in another order, eps = 2^-52), <= 8 sqrt(n) ULP of the
    reference's sum (empirical: rounding errors walk randomly), and |gpu - EXACT sum| <= (sqrt(n) + 8) ULP, the exact sums
    computed in integer arithmetic (oracle.ExactGroupSums).  COUNT is exact.  The observed maxima are printed (pytest -s)."""
    with ThreadPoolExecutor(2) as pool:  # (the exact sums: ~40 s of numpy per table, next to the GPU and oracle runs)
        truths = {"uniform_filtered": pool.submit(_exact_sums_uniform, (0.2, 0.4), _rows("uniform_filtered")),
                  "uniform_all": pool.submit(_exact_sums_uniform, None, _rows("uniform_all"))}
        for name, what in (("uniform_filtered", "uniform v, filtered"), ("uniform_all", "uniform v, no filter")):
            got = _gpu(name)
            want = _oracle_result(name)[2]
            gk, (gs, gc) = _sorted_columns(got)
            wk, (ws, wc) = _sorted_columns(want)
            _assert_keys_equal(gk, wk, what)
            assert np.array_equal(gc, wc), f"{what}: COUNT differs"
            truth = truths[name].result()[wk]
            stats = oracle.check_float_sums(gs, ws, wc, ws, truth=truth, what=what)  # v >= 0: sum|v| == the sum itself
            print(f"\n{what}: max |gpu - reference| = {stats['max_ulp_vs_reference']:.1f} ULP = {stats['max_over_sqrt_n']:.2f} sqrt(n); "
                  f"|gpu - exact| <= {stats['max_ulp_vs_exact']:.1f} ULP, |reference - exact| <= {stats['reference_max_ulp_vs_exact']:.1f} ULP; "
                  f"rows per group up to {int(wc.max())}")


def test_wide_int64_keys_per_group_vs_oracle_at_2_28_rows():
    """Round 5: keys that are NOT small integers at the benchmark's size -- 10^6 distinct Int64 keys spread over the whole type (none
    below 2^32, half negative).  The calibration slice sees them, so the stream never enters narrow mode: pass 1 routes 16-byte rows
    {key, operand} (the ring kernel with 8-row chunks), pass 2 compares 64-bit keys.  With the headline's filter (SUM + COUNT, generic
    row width) and as config 3 is written (no filter, one aggregate: the lean pass 2); every group bit for bit."""
    for name in ("wide_filtered", "wide_all"):
        got = _gpu(name)
        _secs, kept, want = _oracle_result(name)
        assert got.num_rows == 1000000
        assert np.abs(got.column(0).to_numpy().astype(np.float64)).min() >= float(1 << 32)  # no key of the result has a 32-bit form
        _assert_bit_exact(got, want, name + " 2^28")


def test_zipf_keys_per_group_vs_oracle_at_2_28_rows():
    """Skewed keys (SURVEY 8(d)'s Zipf s = 1.0 variant; the generator is log-uniform, p(k) ~ 1/k: keys 0 and 1 own ~5 %
    of the rows each, the 20 hottest keys a quarter), exact values: per group bit-exact; with and without the filter."""
    for name in ("zipf_filtered", "zipf_all"):
        got = _gpu(name)
        _assert_bit_exact(got, _oracle_result(name)[2], name + " 2^28")


@pytest.mark.parametrize("fast", [1, 0])
def test_config2_filter_as_written_mask_and_compaction_at_2_27_row_batches(fast):
    """BASELINE config 2 as written: FilterRelation (filter.rs:46-110) over lat = 49 + 10 u, WHERE lat > 51 AND lat < 53,
    2^28 rows in 2^27-row batches.  The bitmap of every batch and the compacted column are compared with the oracle's
    (orc_filter_next over 1024-row batches) bit for bit; scan.fast = 1 is the static signature, 0 the SSA interpreter."""
    ex.set_option("scan.fast", fast)
    secs, kept, want_cols, want_mask = _oracle_result("cfg2_filter")
    want = want_cols[0]
    t = ex.DeviceTable.synth(SYN_LAT, SEED2, 0, N2)
    rel = ex.FilterRelation(t.scan(BATCH2), ex.compile_scalar_expr(None, PRED2, SCHEMA2), SCHEMA2)
    rel.keep_mask()
    at = 0
    row0 = 0
    while True:
        b = rel.next()
        if b is None:
            break
        bits, rows = rel.last_mask(BATCH2)
        assert rows == min(BATCH2, N2 - row0)
        assert row0 % 8 == 0 and np.array_equal(bits, want_mask[row0 // 8:(row0 + rows + 7) // 8]), f"bitmap of the batch at row {row0} differs"
        got = b.column(0).to_numpy()
        assert b.column(0).null_count == 0
        assert np.array_equal(got.view(np.uint64), want[at:at + len(got)].view(np.uint64)), f"compacted rows of the batch at row {row0} differ"
        assert len(got) == int(np.unpackbits(bits, bitorder="little")[:rows].sum())
        at += len(got)
        row0 += rows
    assert row0 == N2 and at == kept and abs(kept / N2 - 0.2) < 1e-3


def _q1_gpu(syn, batch):
    t = ex.DeviceTable.synth(syn, SEED5, 0, N5)
    return gpu_aggregate([Column(0), Column(1)], AGGS5, SCHEMA5, [], filter_expr=PRED5, source=t.scan(batch))


def _q1_sorted(batch, n_aggr):
    key = batch.column(0).to_numpy() * 2 + batch.column(1).to_numpy()
    order = np.argsort(key, kind="stable")
    return key[order], [batch.column(2 + i).to_numpy()[order] for i in range(n_aggr)]


def test_config5_q1_shape_exact_variant_bit_for_bit():
    """BASELINE config 5's shape through the synth generator, 2^27 rows in 2^26-row batches, exact-arithmetic columns: the
    four SUMs of every group equal the oracle's bit for bit (aggregate.rs:787-952 + update_accumulators :548-612)."""
    got = _q1_gpu(SYN_Q1_EXACT, 1 << 26)
    _secs, kept, want = _oracle_result("q1_exact")
    gk, gv = _q1_sorted(got, 4)
    wk, wv = _q1_sorted(want, 5)
    assert len(gk) == 6 and np.array_equal(gk, wk)
    assert int(wv[4].sum()) == kept
    for i in range(4):
        assert np.array_equal(gv[i].view(np.uint64), wv[i].view(np.uint64)), f"Q1 exact: SUM #{i} differs: {gv[i]} vs {wv[i]}"


def _exact_sums_q1(n_rows):
    """correctly rounded EXACT sums of the four Q1 aggregates per (rf, ls) group (oracle.exact_sums_q1: the arguments as the reference
    computes them, added in integer arithmetic)"""
    return oracle.exact_sums_q1(SYN_Q1_UNIFORM, SEED5, 0, n_rows)[0]


def test_config5_q1_shape_uniform_variant_within_tolerance():
    """bench.py's columns (uniform doubles), one 2^27-row batch.  A parallel sum cannot reproduce the reference's sequential
    rounding: per group with n rows the three bounds of check_float_sums -- n * eps * sum|v| (every term
    is positive, so sum|v| is the sum), 8 sqrt(n) ULP of the reference's sum, and (sqrt(n) + 8) ULP of the EXACT sum
    (oracle.ExactGroupSums over the arguments as the reference computes them).  The observed distances are printed (pytest -s)."""
    with ThreadPoolExecutor(1) as pool:
        truth_f = pool.submit(_exact_sums_q1, N5)
        got = _q1_gpu(SYN_Q1_UNIFORM, 1 << 27)
        want = _oracle_result("q1_uniform")[2]
        gk, gv = _q1_sorted(got, 4)
        wk, wv = _q1_sorted(want, 5)
        assert len(gk) == 6 and np.array_equal(gk, wk)
        n = wv[4].astype(np.float64)
        truth = truth_f.result()
    for i in range(4):
        stats = oracle.check_float_sums(gv[i], wv[i], n, wv[i], truth=truth[i][wk], what=f"Q1 uniform SUM #{i}")
        print(f"\nQ1 uniform SUM #{i}: |gpu - reference| <= {stats['max_ulp_vs_reference']:.0f} ULP = {stats['max_over_sqrt_n']:.2f} sqrt(n); "
              f"|gpu - exact| <= {stats['max_ulp_vs_exact']:.1f} ULP, |reference - exact| <= {stats['reference_max_ulp_vs_exact']:.0f} ULP; groups of up to {int(n.max())} rows")
