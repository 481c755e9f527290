"""Randomised differential tests: random well-typed expression trees over random batches (all 10 numeric types, nulls,
NaN / inf / extreme integers) through Project, Filter and Aggregate on the device against the oracle.  The targeted
tests pin the reference's rules one by one; this one looks for combinations nobody thought of.  Seeds are fixed.

NaN results are compared as "is NaN": the sign / payload of a NaN produced by an operation (inf - inf ...) is
hardware-defined (x86 SSE yields the negative "real indefinite", CDNA the positive canonical NaN) and no reference test
observes it.  MIN / MAX over a group that holds both zeros may return either zero (arrival order in the reference,
total order on the device).  Everything else is bit for bit."""
import os
import sys

import numpy as np
import pyarrow as pa
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle  # noqa: E402
from gpu_util import bits, gpu_aggregate, gpu_filter, gpu_project  # noqa: E402

from datafusion_archive_amd import execution as ex  # noqa: E402
from datafusion_archive_amd.logicalplan import (AggregateFunction, BinaryExpr, Cast, Column, DataType, Literal, Operator,  # noqa: E402
                                                ScalarValue)

pytestmark = pytest.mark.gpu

NP_TYPES = [(DataType.Float64, np.float64), (DataType.Float32, np.float32), (DataType.Int64, np.int64), (DataType.Int32, np.int32),
            (DataType.Int16, np.int16), (DataType.Int8, np.int8), (DataType.UInt64, np.uint64), (DataType.UInt32, np.uint32),
            (DataType.UInt16, np.uint16), (DataType.UInt8, np.uint8)]
LIT = {DataType.Float64: ScalarValue.Float64, DataType.Float32: ScalarValue.Float32, DataType.Int64: ScalarValue.Int64,
       DataType.Int32: ScalarValue.Int32, DataType.Int16: ScalarValue.Int16, DataType.Int8: ScalarValue.Int8,
       DataType.UInt64: ScalarValue.UInt64, DataType.UInt32: ScalarValue.UInt32, DataType.UInt16: ScalarValue.UInt16,
       DataType.UInt8: ScalarValue.UInt8}
CMP = [Operator.Eq, Operator.NotEq, Operator.Lt, Operator.LtEq, Operator.Gt, Operator.GtEq]


def random_batch(rng, n, with_nulls):
    arrays, names = [], []
    for dt, npt in NP_TYPES:
        if np.issubdtype(npt, np.floating):
            a = (rng.standard_normal(n) * 10.0 ** rng.integers(-3, 4, n)).astype(npt)
            special = rng.random(n)
            a[special < 0.02] = np.nan
            a[(special >= 0.02) & (special < 0.04)] = np.inf
            a[(special >= 0.04) & (special < 0.06)] = -np.inf
            a[(special >= 0.06) & (special < 0.10)] = 0.0
            a[(special >= 0.10) & (special < 0.12)] = -0.0
        else:
            info = np.iinfo(npt)
            a = rng.integers(max(info.min, -50), min(info.max, 50), n, endpoint=True).astype(npt)
            special = rng.random(n)
            a[special < 0.03] = info.max
            a[(special >= 0.03) & (special < 0.06)] = info.min
        arr = pa.array(a, mask=(rng.random(n) < 0.15)) if with_nulls else pa.array(a)
        arrays.append(arr)
        names.append(dt.name.lower())
    arrays.append(pa.array([("g%d" % int(x)) * int(x % 3) for x in rng.integers(0, 12, n)]))  # column 10: Utf8 (never an operand)
    names.append("txt")
    b = pa.RecordBatch.from_arrays(arrays, names=names)
    if n > 8 and rng.random() < 0.5:  # a slice: non-zero Arrow offsets (values, validity bit offsets, Utf8 offsets)
        lo = int(rng.integers(1, min(n - 1, 70)))
        b = b.slice(lo, n - lo - int(rng.integers(0, 3)))
    return b


class Gen:
    def __init__(self, rng, allow_divide):
        self.rng = rng
        self.allow_divide = allow_divide

    def literal(self, t):
        dt, npt = NP_TYPES[t]
        if np.issubdtype(npt, np.floating):
            v = float(self.rng.choice([0.5, -1.25, 3.0, 1e-3, 7.0, -2.0]))
        else:
            info = np.iinfo(npt)
            v = int(self.rng.choice([1, 2, 3, 7, min(info.max, 100), max(info.min, -3)]))
        return Literal(LIT[dt](v))

    def numeric(self, t, depth):
        r = self.rng.random()
        if depth <= 0 or r < 0.3:
            return Column(t) if self.rng.random() < 0.8 else self.literal(t)
        if r < 0.45:  # cast from another type
            src = int(self.rng.integers(0, len(NP_TYPES)))
            return Cast(self.numeric(src, depth - 1), NP_TYPES[t][0])
        ops = [Operator.Plus, Operator.Minus, Operator.Multiply] + ([Operator.Divide] if self.allow_divide else [])
        op = ops[int(self.rng.integers(0, len(ops)))]
        right = self.numeric(t, depth - 1)
        if op == Operator.Divide and self.rng.random() < 0.7:
            right = self.literal(t)  # mostly safe divisors; the rest exercises DivideByZero
        return BinaryExpr(self.numeric(t, depth - 1), op, right)

    def boolean(self, depth):
        if depth > 0 and self.rng.random() < 0.4:
            return BinaryExpr(self.boolean(depth - 1), Operator.And if self.rng.random() < 0.5 else Operator.Or, self.boolean(depth - 1))
        t = int(self.rng.integers(0, len(NP_TYPES)))
        return BinaryExpr(self.numeric(t, 1), CMP[int(self.rng.integers(0, len(CMP)))], self.numeric(t, 1))


def canon(arr):
    """bit patterns with every NaN mapped to one token"""
    out = bits(arr)
    if pa.types.is_floating(arr.type):
        vals = arr.to_pylist()
        out = ["nan" if (v is not None and v != v) else b for v, b in zip(vals, out)]
    return out


def same_batches(got, want, what):
    assert got.num_columns == want.num_columns and got.num_rows == want.num_rows, what
    for c in range(got.num_columns):
        assert got.column(c).type == want.column(c).type, (what, c)
        g, w = canon(got.column(c)), canon(want.column(c))
        if g != w:
            i = [j for j, (a, b) in enumerate(zip(g, w)) if a != b][0]
            raise AssertionError(f"{what}: column {c} row {i}: got {got.column(c)[i]} want {want.column(c)[i]}")


def run_both(dev, ora, what):
    """Both succeed with equal results, or both fail with the same error class. NotImplemented '... more than ...' = the fused program
    limits of the device (kMaxRegs / kMaxCols / kMaxImm) hit by ONE expression tree (aggregates beyond them are chunked): skipped."""
    try:
        want = ora()
    except oracle.OracleError as e:
        with pytest.raises(ex.ExecutionError) as ei:
            dev()
        if "more than" in ei.value.message and ei.value.kind == "NotImplemented":
            return "skipped"
        assert ei.value.code == e.code, (what, str(e), ei.value.message)
        return "error"
    try:
        got = dev()
    except ex.ExecutionError as e:
        if "more than" in e.message and e.kind == "NotImplemented":
            return "skipped"
        raise AssertionError(f"{what}: device failed with {e}, oracle succeeded")
    return got, want


@pytest.mark.parametrize("with_nulls", [False, True])
@pytest.mark.parametrize("fast", [1, 0])
def test_fuzz_project(with_nulls, fast):
    ex.set_option("scan.fast", fast)
    rng = np.random.default_rng(4000 + with_nulls)
    stats = {"ok": 0, "error": 0, "skipped": 0}
    for case in range(120):
        b = random_batch(rng, int(rng.integers(1, 3000)), with_nulls)
        g = Gen(rng, allow_divide=True)
        exprs = [g.numeric(int(rng.integers(0, len(NP_TYPES))), int(rng.integers(1, 4))) for _ in range(int(rng.integers(1, 4)))]
        if rng.random() < 0.5:
            exprs.append(g.boolean(1))
        if rng.random() < 0.3:
            exprs.append(Column(10))
        r = run_both(lambda: gpu_project(exprs, b.schema, [b])[0], lambda: oracle.project_next(exprs, b), f"project case {case}: {exprs}")
        if isinstance(r, tuple):
            same_batches(r[0], r[1], f"project case {case}: {exprs}")
            stats["ok"] += 1
        else:
            stats[r] += 1
    print(f"fuzz project nulls={with_nulls} fast={fast}: {stats}")
    assert stats["ok"] >= 60


@pytest.mark.parametrize("with_nulls", [False, True])
@pytest.mark.parametrize("fast", [1, 0])
def test_fuzz_filter(with_nulls, fast):
    ex.set_option("scan.fast", fast)
    rng = np.random.default_rng(5000 + with_nulls)
    stats = {"ok": 0, "error": 0, "skipped": 0}
    for case in range(120):
        b = random_batch(rng, int(rng.integers(1, 5000)), with_nulls)
        pred = Gen(rng, allow_divide=False).boolean(int(rng.integers(0, 3)))
        r = run_both(lambda: gpu_filter(pred, b.schema, [b])[0], lambda: oracle.filter_next(pred, b), f"filter case {case}: {pred}")
        if isinstance(r, tuple):
            same_batches(r[0], r[1], f"filter case {case}: {pred}")
            stats["ok"] += 1
        else:
            stats[r] += 1
    print(f"fuzz filter nulls={with_nulls} fast={fast}: {stats}")
    assert stats["ok"] >= 80


@pytest.mark.parametrize("strategy", [0, 1, 3])
@pytest.mark.parametrize("with_nulls", [False, True])
def test_fuzz_aggregate(with_nulls, strategy):
    """GROUP BY 0..2 integer columns; MIN / MAX / COUNT over any expression, SUM over integer expressions (wrapping: order
    independent) -- float SUM needs the exact-data tests."""
    ex.set_option("agg.strategy", strategy)
    rng = np.random.default_rng(6000 + with_nulls + 10 * strategy)
    stats = {"ok": 0, "error": 0, "skipped": 0}
    int_types = [i for i, (_dt, npt) in enumerate(NP_TYPES) if not np.issubdtype(npt, np.floating)]
    for case in range(80):
        batches = [random_batch(rng, int(rng.integers(1, 4000)), with_nulls) for _ in range(int(rng.integers(1, 4)))]
        schema = batches[0].schema
        g = Gen(rng, allow_divide=False)
        n_keys = int(rng.integers(0, 3)) if strategy != 3 else int(rng.integers(0, 2))
        keys = [Column(int(rng.choice(int_types))) for _ in range(n_keys)]
        if keys and strategy != 3 and rng.random() < 0.25:
            keys[0] = Column(10)  # a Utf8 key (device dictionary)
        if strategy == 3:  # the partitioned strategy's flavours, at random: 12-byte rows, hot-key pairs, deferred pass 2, layouts
            flavour = {"agg.narrow_keys": int(rng.choice([-1, 1])), "agg.hot_keys": int(rng.choice([0, 1])),
                       "agg.partition_defer": int(rng.choice([1, 4])), "agg.partition_layout": int(rng.integers(0, 3)),
                       "agg.pass2_stream": int(rng.choice([0, 1])), "agg.ctrl_snapshot": int(rng.choice([0, 1])),
                       "agg.narrow_chunk16": int(rng.choice([0, 1]))}
            for k, v in flavour.items():
                ex.set_option(k, v)
        aggs = []
        n_aggs = int(rng.integers(1, 5)) if rng.random() < 0.85 else int(rng.integers(5, 13))  # sometimes more than 8 accumulators: chunks
        if strategy == 3 and rng.random() < 0.5:
            n_aggs = 1  # the one-aggregate kernels (narrow rows, lean pass 2) are the partitioned strategy's main line
        for _ in range(n_aggs):
            t = int(rng.integers(0, len(NP_TYPES)))
            fn = str(rng.choice(["min", "max", "count", "sum", "avg"]))
            if fn in ("sum", "avg"):
                t = int(rng.choice(int_types))
            rt = DataType.UInt64 if fn == "count" else NP_TYPES[t][0]
            aggs.append(AggregateFunction(fn, [g.numeric(t, int(rng.integers(0, 3)))], rt))
        pred = g.boolean(1) if rng.random() < 0.5 else None
        src = batches if pred is None else None

        def ora():
            inp = batches if pred is None else [oracle.filter_next(pred, b) for b in batches]
            return oracle.aggregate(keys, aggs, inp)
        r = run_both(lambda: gpu_aggregate(keys, aggs, schema, batches, filter_expr=pred), ora, f"aggregate case {case}: {keys} {aggs} where {pred}")
        if isinstance(r, tuple):
            got, want = r
            if n_keys == 0:
                same_batches(got, want, f"aggregate case {case}")
            else:
                def as_dict(batch):
                    # MIN / MAX of a group holding both -0.0 and +0.0: the reference's f64::min / max keeps whichever
                    # arrived in the position its x86 lowering favours, the device's total order says -0.0 < +0.0; the
                    # two zeros are numerically equal, so aggregates of floats are compared by VALUE (NaN == NaN)
                    cols = [canon(batch.column(i)) for i in range(n_keys)]
                    for i in range(n_keys, batch.num_columns):
                        col = batch.column(i)
                        if pa.types.is_floating(col.type):
                            cols.append(["nan" if (v is not None and v != v) else (0.0 if v == 0 else v) for v in col.to_pylist()])
                        else:
                            cols.append(canon(col))
                    return {tuple(c[r] for c in cols[:n_keys]): tuple(c[r] for c in cols[n_keys:]) for r in range(batch.num_rows)}
                gd, wd = as_dict(got), as_dict(want)
                assert len(gd) == got.num_rows and set(gd) == set(wd), f"aggregate case {case}: group sets differ"
                bad = [k for k in wd if gd[k] != wd[k]]
                assert not bad, f"aggregate case {case}: {keys} {aggs} where {pred}: group {bad[0]} got {gd[bad[0]]} want {wd[bad[0]]}"
            stats["ok"] += 1
        else:
            stats[r] += 1
        del src
    for k, v in (("agg.narrow_keys", -1), ("agg.hot_keys", -1), ("agg.partition_defer", 0), ("agg.partition_layout", 1),
                 ("agg.pass2_stream", 1), ("agg.ctrl_snapshot", 1), ("agg.narrow_chunk16", 1)):
        ex.set_option(k, v)
    print(f"fuzz aggregate nulls={with_nulls} strategy={strategy}: {stats}")
    assert stats["ok"] >= 40


@pytest.mark.parametrize("fewgroup", [1, 0])
def test_fuzz_few_groups_many_batches(fewgroup):
    """<= 6 groups over 3..6 batches: from the second batch on the register-accumulator kernel (k_fewgroup_agg) runs --
    its generic instances (1 or 2 key words, 1..4 aggregates of any accumulator kind, optional predicate) against the
    oracle, and against the LDS path (agg.fewgroup=0)."""
    ex.set_option("agg.fewgroup", fewgroup)
    rng = np.random.default_rng(7000)
    int_types = [i for i, (_dt, npt) in enumerate(NP_TYPES) if not np.issubdtype(npt, np.floating)]
    done = 0
    for case in range(60):
        n_batches = int(rng.integers(3, 7))
        batches = []
        for _ in range(n_batches):
            b = random_batch(rng, int(rng.integers(200, 5000)), False)
            m = b.num_rows
            g3 = pa.array(rng.integers(0, 3, m).astype(np.int64))
            g2 = pa.array(rng.integers(-1, 1, m).astype(np.int16))
            batches.append(pa.RecordBatch.from_arrays(list(b.columns) + [g3, g2], names=list(b.schema.names) + ["g3", "g2"]))
        schema = batches[0].schema
        keys = [Column(11)] if rng.random() < 0.5 else [Column(11), Column(12)]
        g = Gen(rng, allow_divide=False)
        aggs = []
        for _ in range(int(rng.integers(1, 5))):
            t = int(rng.integers(0, len(NP_TYPES)))
            fn = str(rng.choice(["min", "max", "count", "sum"]))
            if fn == "sum":
                t = int(rng.choice(int_types))
            rt = DataType.UInt64 if fn == "count" else NP_TYPES[t][0]
            arg = Column(t) if rng.random() < 0.7 else g.numeric(t, 1)
            aggs.append(AggregateFunction(fn, [arg], rt))
        pred = None
        if rng.random() < 0.5:
            t = int(rng.integers(0, len(NP_TYPES)))
            pred = BinaryExpr(Column(t), CMP[int(rng.integers(0, len(CMP)))], g.literal(t))
        got = gpu_aggregate(keys, aggs, schema, batches, filter_expr=pred)
        want = oracle.aggregate(keys, aggs, batches if pred is None else [oracle.filter_next(pred, b) for b in batches])

        def as_dict(batch, nk=len(keys)):
            cols = [canon(batch.column(i)) for i in range(nk)]
            for i in range(nk, batch.num_columns):
                col = batch.column(i)
                if pa.types.is_floating(col.type):
                    cols.append(["nan" if (v is not None and v != v) else (0.0 if v == 0 else v) for v in col.to_pylist()])
                else:
                    cols.append(canon(col))
            return {tuple(c[r] for c in cols[:nk]): tuple(c[r] for c in cols[nk:]) for r in range(batch.num_rows)}
        gd, wd = as_dict(got), as_dict(want)
        assert gd == wd, f"few groups case {case}: {keys} {aggs} where {pred}: got {gd} want {wd}"
        done += 1
    assert done == 60
