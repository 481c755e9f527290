"""Independent cross-check of the oracle on behaviour the reference's tests do not pin (pyarrow.compute
and numpy are available).  On NULL-FREE data arrow 0.12's kernels, today's pyarrow and numpy's IEEE / wrapping arithmetic
must all agree, so the oracle is compared with them: comparisons, AND / OR, + - * (floats and wrapping integers), / on
floats, casts, stream compaction, grouped and ungrouped MIN / MAX / COUNT / integer SUM.  (Null handling and float SUM
order are arrow-0.12 specific and stay "unpinned".)"""
import os
import sys

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle  # noqa: E402

from datafusion_archive_amd.logicalplan import AggregateFunction, BinaryExpr, Cast, Column, DataType, Operator  # noqa: E402

TYPES = [(np.float64, DataType.Float64), (np.float32, DataType.Float32), (np.int64, DataType.Int64), (np.int32, DataType.Int32),
         (np.int8, DataType.Int8), (np.uint64, DataType.UInt64), (np.uint16, DataType.UInt16)]


def batch(rng, n):
    cols = []
    for npt, _ in TYPES:
        for _rep in range(2):
            if np.issubdtype(npt, np.floating):
                cols.append(pa.array((rng.standard_normal(n) * 100).astype(npt)))
            else:
                info = np.iinfo(npt)
                cols.append(pa.array(rng.integers(max(info.min, -1000), min(info.max, 1000), n, endpoint=True).astype(npt)))
    return pa.RecordBatch.from_arrays(cols, names=[f"c{i}" for i in range(len(cols))])


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_oracle_elementwise_ops_agree_with_numpy(seed):
    rng = np.random.default_rng(seed)
    b = batch(rng, 3000)
    for t, (npt, _dt) in enumerate(TYPES):
        x, y = b.column(2 * t).to_numpy(), b.column(2 * t + 1).to_numpy()
        cx, cy = Column(2 * t), Column(2 * t + 1)
        for op, fn in [(Operator.Eq, np.equal), (Operator.NotEq, np.not_equal), (Operator.Lt, np.less), (Operator.LtEq, np.less_equal),
                       (Operator.Gt, np.greater), (Operator.GtEq, np.greater_equal)]:
            got = oracle.eval_expr(BinaryExpr(cx, op, cy), b).to_numpy(zero_copy_only=False)
            assert np.array_equal(got, fn(x, y)), (npt, op)
        with np.errstate(over="ignore"):
            for op, fn in [(Operator.Plus, np.add), (Operator.Minus, np.subtract), (Operator.Multiply, np.multiply)]:
                got = oracle.eval_expr(BinaryExpr(cx, op, cy), b).to_numpy()
                want = fn(x, y)  # numpy: IEEE for floats, wrapping for integers -- Rust release semantics
                assert got.dtype == want.dtype and np.array_equal(got.view(np.uint8), want.view(np.uint8)), (npt, op)
        if np.issubdtype(npt, np.floating):
            got = oracle.eval_expr(BinaryExpr(cx, Operator.Divide, Column(2 * t + 1)), b).to_numpy()
            with np.errstate(divide="ignore", invalid="ignore"):
                assert np.array_equal(got.view(np.uint8), (x / y).view(np.uint8))
    lt = BinaryExpr(Column(0), Operator.Lt, Column(1))
    gt = BinaryExpr(Column(4), Operator.Gt, Column(5))
    x0, x1, x4, x5 = (b.column(i).to_numpy() for i in (0, 1, 4, 5))
    assert np.array_equal(oracle.eval_expr(BinaryExpr(lt, Operator.And, gt), b).to_numpy(zero_copy_only=False), (x0 < x1) & (x4 > x5))
    assert np.array_equal(oracle.eval_expr(BinaryExpr(lt, Operator.Or, gt), b).to_numpy(zero_copy_only=False), (x0 < x1) | (x4 > x5))
    # casts with Rust `as` semantics where numpy agrees: int -> float, float -> wider float, int widening
    assert np.array_equal(oracle.eval_expr(Cast(Column(4), DataType.Float64), b).to_numpy(), x4.astype(np.float64))
    assert np.array_equal(oracle.eval_expr(Cast(Column(2), DataType.Float64), b).to_numpy(), b.column(2).to_numpy().astype(np.float64))
    assert np.array_equal(oracle.eval_expr(Cast(Column(6), DataType.Int64), b).to_numpy(), b.column(6).to_numpy().astype(np.int64))
    assert np.array_equal(oracle.eval_expr(Cast(Column(0), DataType.Int32), b).to_numpy(), np.trunc(x0).astype(np.int32))


@pytest.mark.parametrize("seed", [4, 5])
def test_oracle_filter_and_aggregates_agree_with_pyarrow(seed):
    rng = np.random.default_rng(seed)
    b = batch(rng, 5000)
    pred = BinaryExpr(Column(0), Operator.Gt, Column(1))
    got = oracle.filter_next(pred, b)
    want = pa.Table.from_batches([b]).filter(pc.greater(b.column(0), b.column(1))).combine_chunks().to_batches()[0]
    assert got.equals(want)
    # grouped by an int16-ish key (column 12: uint16 in [0, 1000]); MIN / MAX of f64, COUNT, SUM of int64
    key = pa.array((b.column(12).to_numpy() % 37).astype(np.uint16))
    bb = pa.RecordBatch.from_arrays([key, b.column(0), b.column(4)], names=["k", "v", "i"])
    aggs = [AggregateFunction("min", [Column(1)], DataType.Float64), AggregateFunction("max", [Column(1)], DataType.Float64),
            AggregateFunction("count", [Column(1)], DataType.UInt64), AggregateFunction("sum", [Column(2)], DataType.Int64)]
    got = oracle.aggregate([Column(0)], aggs, [bb.slice(0, 2000), bb.slice(2000)])
    want = pa.Table.from_batches([bb]).group_by("k").aggregate([("v", "min"), ("v", "max"), ("v", "count"), ("i", "sum")])
    g = {r[0]: tuple(r[1:]) for r in zip(*[got.column(i).to_pylist() for i in range(5)])}
    w = {k: (a, c, d, e) for k, a, c, d, e in zip(*[want.column(n).to_pylist() for n in ("k", "v_min", "v_max", "v_count", "i_sum")])}
    assert g == w
    tot = oracle.aggregate([], aggs, [bb])
    v, i = bb.column(1).to_numpy(), bb.column(2).to_numpy()
    assert tot.to_pylist()[0] == {"c0": float(v.min()), "c1": float(v.max()), "c2": len(v), "c3": int(i.sum())}


def test_oracle_sort_agrees_with_pyarrow_and_limit():
    """ORDER BY semantics the library defines (reference: unimplemented): on null-free, NaN-free data the naive oracle must
    equal pyarrow's stable sort; NULL placement (largest) is checked against pyarrow's null_placement."""
    rng = np.random.default_rng(11)
    n = 4000
    b = pa.RecordBatch.from_arrays([pa.array(rng.integers(-5, 5, n).astype(np.int64)), pa.array(rng.standard_normal(n)),
                                    pa.array([f"s{int(x)}" for x in rng.integers(0, 50, n)]), pa.array(np.arange(n))],
                                   names=["a", "f", "s", "row"])
    t = pa.Table.from_batches([b])
    for keys, pa_keys in [([(Column(0), True), (Column(1), False)], [("a", "ascending"), ("f", "descending")]),
                          ([(Column(2), False), (Column(0), True)], [("s", "descending"), ("a", "ascending")]),
                          ([(Column(1), True)], [("f", "ascending")])]:
        got = oracle.sort_batches([b.slice(0, 1500), b.slice(1500)], keys)
        want = t.take(pc.sort_indices(t, sort_keys=pa_keys))  # pyarrow's sort is stable
        assert got.column(3).to_pylist() == want.column("row").to_pylist(), keys
    bn = pa.RecordBatch.from_arrays([pa.array(rng.integers(0, 9, n).astype(np.int64), mask=rng.random(n) < 0.2), pa.array(np.arange(n))], names=["a", "row"])
    tn = pa.Table.from_batches([bn])
    asc = oracle.sort_batches([bn], [(Column(0), True)])
    assert asc.column(1).to_pylist() == tn.take(pc.sort_indices(tn, sort_keys=[("a", "ascending")], null_placement="at_end")).column("row").to_pylist()
    desc = oracle.sort_batches([bn], [(Column(0), False)])
    assert desc.column(1).to_pylist() == tn.take(pc.sort_indices(tn, sort_keys=[("a", "descending")], null_placement="at_start")).column("row").to_pylist()
    assert [x.num_rows for x in oracle.limit_batches([bn.slice(0, 10), bn.slice(10, 10)], 13)] == [10, 3]
