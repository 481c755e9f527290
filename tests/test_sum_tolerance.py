"""The Float64 SUM tolerance (tests/oracle.py: check_float_sums, ExactGroupSums) on the CPU: the exact
sums are exact, the reference's sequential sums sit where the random-walk model says, and a result that LOST ONE ROW of one
group -- the failure a loose tolerance would wave through -- trips every bound."""
import math

import numpy as np
import pytest

import oracle
from datafusion_archive_amd.logicalplan import AggregateFunction, Column, DataType

SUM_V = AggregateFunction("SUM", [Column(1)], DataType.Float64)
COUNT_V = AggregateFunction("COUNT", [Column(1)], DataType.UInt64)


def test_exact_group_sums_are_exact():
    rng = np.random.default_rng(1)
    n, g = 200000, 37
    keys = rng.integers(0, g, n)
    vals = rng.integers(0, 1 << 53, n, dtype=np.uint64).astype(np.float64) * 2.0 ** -53  # any point of the 2^-53 grid
    ex_ = oracle.ExactGroupSums(g)
    for lo in range(0, n, 50000):
        ex_.add(keys[lo:lo + 50000], vals[lo:lo + 50000])
    truth = ex_.result()
    for i in range(g):
        assert truth[i] == math.fsum(vals[keys == i].tolist())  # Shewchuk: the correctly rounded exact sum
    # another fixed-point window: products of bounded factors (the Q1 shape's arguments), scale 52
    price = 900.0 + 104100.0 * rng.random(n)
    disc = 0.10 * rng.random(n)
    arg = price * (1.0 - disc)
    ex2 = oracle.ExactGroupSums(g, scale=52)
    ex2.add(keys, arg)
    t2 = ex2.result()
    for i in range(g):
        assert t2[i] == math.fsum(arg[keys == i].tolist())


def _reference_and_truth(n_rows, groups):
    syn = [("k", oracle.SYNTH_I64_UNIFORM, 0, float(groups), 0.0), ("v", oracle.SYNTH_F64_UNIFORM, 1, 0.0, 1.0)]
    _s, _kept, ref = oracle.run_synth_query(syn, 0xDF02, 0, n_rows, 1024, None, [Column(0)], [SUM_V, COUNT_V])
    k = ref.column(0).to_numpy()
    order = np.argsort(k)
    ref_sum, cnt = ref.column(1).to_numpy()[order], ref.column(2).to_numpy()[order].astype(np.float64)
    keys = oracle.synth_column(oracle.SYNTH_I64_UNIFORM, 0, float(groups), 0.0, 0xDF02, 0, n_rows)
    vals = oracle.synth_column(oracle.SYNTH_F64_UNIFORM, 1, 0.0, 1.0, 0xDF02, 0, n_rows)
    ex_ = oracle.ExactGroupSums(groups)
    ex_.add(keys, vals)
    return keys, vals, ref_sum, cnt, ex_.result()


def test_reference_sums_sit_within_the_random_walk_bound_and_the_exact_sums_pass_every_check():
    keys, vals, ref_sum, cnt, truth = _reference_and_truth(2000000, 7)   # ~286 000 rows per group
    stats = oracle.check_float_sums(truth, ref_sum, cnt, ref_sum, truth=truth, what="exact sums as the product's result")
    assert 0.0 < stats["max_over_sqrt_n"] < 4.0, stats    # the reference's own rounding: a fraction of sqrt(n) ULP
    assert stats["max_ulp_vs_exact"] == 0.0


@pytest.mark.parametrize("groups,rows", [(7, 2000000), (1000, 300000), (1, 10000000)])  # (the last: ONE lost row of a 10^7-row group)
def test_a_result_that_lost_one_row_trips_the_tolerance(groups, rows):
    keys, vals, ref_sum, cnt, truth = _reference_and_truth(rows, groups)
    victim = int(keys[12345])
    lost = truth.copy()
    lost[victim] = truth[victim] - vals[12345]   # a "device result" that dropped row 12345 (exact otherwise)
    with pytest.raises(AssertionError):
        oracle.check_float_sums(lost, ref_sum, cnt, ref_sum, what="one row lost")
    with pytest.raises(AssertionError):
        oracle.check_float_sums(lost, ref_sum, cnt, ref_sum, truth=truth, what="one row lost")
    # an error the PROVEN bound alone accepts (n * eps * sum|v| is about n / 2 ULP) and the empirical one does not: a
    # result n / 4 ULP off -- e.g. a kernel that added a row's LOW bits wrongly throughout
    off = truth.copy()
    off[victim] = truth[victim] + 0.25 * cnt[victim] * np.spacing(truth[victim])
    if 0.25 * cnt[victim] > 2.0 * oracle.WORKING_SQRT_N_ULP * math.sqrt(cnt[victim]):  # (groups of more than 2^12 rows: the bounds differ by more than 2)
        assert abs(off[victim] - ref_sum[victim]) <= cnt[victim] * 2.0 ** -52 * ref_sum[victim]
        with pytest.raises(AssertionError, match="8 sqrt"):
            oracle.check_float_sums(off, ref_sum, cnt, ref_sum, what="n / 4 ULP off")
    # ... and so does one row in ten thousand of EVERY group, whatever the group size
    drop = np.zeros(groups)
    idx = np.arange(0, rows, 10000)
    np.add.at(drop, keys[idx], vals[idx])
    with pytest.raises(AssertionError):
        oracle.check_float_sums(truth - drop, ref_sum, cnt, ref_sum, what="1 row in 10^4 lost")
