#!/usr/bin/env python3
"""Soak run of the lock-free device protocols (single-pass filter look-back, LDS rings, wave-specialised pass 1, deferred pass 2)
over random table sizes, batch widths, selectivities and group counts: every iteration checks invariants that hold for the
EXACT data distribution (every partial sum is representable, so any order of additions gives the same bits):

  rows kept by FilterRelation           == COUNT of the fused predicate + reduce
  SUM over FilterRelation's output      == SUM of the fused predicate + reduce          (bit for bit)
  sum over groups of SUM(v)             == that same SUM                                  (bit for bit)
  sum over groups of COUNT(v)           == rows kept;  number of groups <= the key range
  a second run of the same query        == the first                                     (every group, bit for bit)
  the same query under ANOTHER kernel family (interpreter, global table, ring kernel, two-pass filter, wide rows ...) with a
  random aggregate set (SUM / COUNT / MIN / MAX / AVG of v, of a second column w, of v * c) and predicate shape (two-sided,
  one-sided, three terms, none)        == under the defaults                            (every group, every aggregate, bit for bit)

usage: soak.py [seconds] [seed] [first iteration]    exit code 1 on the first violation or error (prints the case that failed)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import pyarrow as pa  # noqa: E402
from datafusion_archive_amd import execution as ex  # noqa: E402
from datafusion_archive_amd.logicalplan import (AggregateFunction, BinaryExpr, Column, DataType, Literal, Operator,  # noqa: E402
                                                ScalarValue)

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20260925)
skip_to = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # replay: draw the same cases, run from this iteration on
ex.init(0)
f64 = DataType.Float64
schema = pa.schema([("k", pa.int64()), ("v", pa.float64())])


def lit(v):
    return Literal(ScalarValue.Float64(float(v)))


def by_key(b):
    k = b.column(0).to_numpy()
    o = np.argsort(k, kind="stable")
    return [k[o]] + [b.column(i).to_numpy()[o] for i in range(1, b.num_columns)]


schema3 = pa.schema([("k", pa.int64()), ("v", pa.float64()), ("w", pa.float64())])
# (option, value of the alternative run, default): kernel families and the rarely taken paths (a table that starts tiny and grows
#  by rehash + spill replay; regions that overflow into the spill list; hot-key pairs forced on)
ALTERNATIVES = [("agg.capacity_log2", 9, 0), ("agg.capacity_log2", 14, 0), ("agg.hot_keys", 1, -1), ("scan.fast", 0, 1), ("agg.strategy", 1, 0), ("agg.strategy", 3, 0), ("agg.pass1_ws", 0, 8), ("filter.single_pass", 0, 1),
                ("agg.narrow_keys", 0, -1), ("agg.shared_operand", 0, 1), ("agg.merge_scan_batches", 0, 1), ("agg.partition_defer", 1, 0),
                ("scan.plan", 0, 1), ("scan.plan", 2, 1), ("agg.split_aggregates", 0, 1),
                ("agg.pair_scan", 0, 1), ("agg.shared_planes", 0, 1)]  # round 6: a scan per aggregate instead of the pair scan; all planes in one block  # round 4: scan plans off / before the signatures; one scan for all aggregates


def aggregate_sets():
    v, w = Column(1), Column(2)
    u64 = DataType.UInt64
    A = AggregateFunction
    return [[A("SUM", [v], f64)], [A("SUM", [v], f64), A("COUNT", [v], u64)], [A("SUM", [v], f64), A("MIN", [v], f64), A("MAX", [v], f64)],
            [A("AVG", [v], f64)], [A("SUM", [v], f64), A("MIN", [w], f64)], [A("MAX", [w], f64), A("COUNT", [w], u64), A("SUM", [v], f64)],
            [A("SUM", [BinaryExpr(v, Operator.Multiply, lit(2.5))], f64)], [A("SUM", [BinaryExpr(v, Operator.Plus, w)], f64)],
            [A("MIN", [v], f64)], [A("COUNT", [v], u64)]]


def same_batches(x, y):
    if x.num_rows != y.num_rows or x.num_columns != y.num_columns:
        return False
    if x.num_rows == 0:
        return True
    a, b = by_key(x), by_key(y)
    return all(np.array_equal(np.ascontiguousarray(p).view(np.uint64) if p.dtype == np.float64 else p,
                              np.ascontiguousarray(q).view(np.uint64) if q.dtype == np.float64 else q) for p, q in zip(a, b))


sum_v = AggregateFunction("SUM", [Column(1)], f64)
count_v = AggregateFunction("COUNT", [Column(1)], DataType.UInt64)
t_end = time.time() + budget
it = 0
while time.time() < t_end:
    it += 1
    n = int(rng.choice([1, 63, 64, 65, 4097, int(rng.integers(1, 1 << 20)), int(rng.integers(1 << 20, 1 << 26)), int(rng.integers(1 << 26, 3 << 27))]))
    groups = int(rng.choice([1, 7, 5000, 20000, 100000, 1000000, 3000000]))  # (3e6 keys outgrow the default 2^21-slot table)
    kind = ex.SYNTH_I64_ZIPF if rng.random() < 0.25 else ex.SYNTH_I64_UNIFORM
    batch = int(rng.choice([1 << 27, 1 << 26, 1 << 24, 1 << 22, (int(rng.integers(1, 1 << 18)) * 64)]))
    a, b = sorted(rng.integers(0, 1 << 20, 2) / 1024.0)
    mode = rng.integers(0, 5)
    lo, hi = [(a, b), (-1.0, 2000.0), (5000.0, 6000.0), (a, a), (0.0, b)][mode]  # some / all / none / empty range / prefix
    ops = [(Operator.Gt, Operator.Lt), (Operator.GtEq, Operator.LtEq), (Operator.GtEq, Operator.Lt), (Operator.Gt, Operator.LtEq)][int(rng.integers(0, 4))]
    pred = BinaryExpr(BinaryExpr(Column(1), ops[0], lit(lo)), Operator.And, BinaryExpr(Column(1), ops[1], lit(hi)))
    seed = int(rng.integers(1, 1 << 30))
    syn = [("k", kind, 0, float(groups), 1.0), ("v", ex.SYNTH_F64_EXACT, 1, 0.0, 0.0)]
    case = f"iteration {it}: n={n} groups={groups} kind={kind} batch={batch} pred=({ops[0].name} {lo}, {ops[1].name} {hi}) seed={seed}"
    if it < skip_to:
        continue
    table = ex.DeviceTable.synth(syn, seed, 0, n)

    def filt():
        return ex.FilterRelation(table.scan(batch), ex.compile_scalar_expr(None, pred, schema), schema)

    def agg(group, aggs):
        rel = ex.AggregateRelation(None, filt(), [ex.compile_scalar_expr(None, g, schema) for g in group], [ex.compile_expr(None, x, schema) for x in aggs])
        out = rel.next()
        assert rel.next() is None
        return out

    def soak_case():
        ref = agg([], [sum_v, count_v])
        want_sum = ref.column(0)[0].as_py()
        want_cnt = ref.column(1)[0].as_py() or 0
        same = lambda x: want_cnt == 0 or np.float64(x).view(np.uint64) == np.float64(want_sum).view(np.uint64)  # noqa: E731
        kept = ex.drain_on_device(filt())[0]
        if kept != want_cnt:
            return False, f"kept {kept} != COUNT {want_cnt}"
        if n <= (1 << 26):  # the compacted output itself (downloaded: bounded)
            got_sum, got_rows = 0.0, 0
            for rb in filt():
                v = rb.column(1).to_numpy()
                got_rows += len(v)
                got_sum += float(np.sum(v))  # exact data: any order
            if got_rows != want_cnt or not same(got_sum):
                return False, f"output rows {got_rows} sum {got_sum!r} != {want_cnt} {want_sum!r}"
        k1 = by_key(agg([Column(0)], [sum_v, count_v]))
        tot = float(np.sum(k1[1])) if len(k1[1]) else 0.0
        if not (int(np.sum(k1[2])) == want_cnt and same(tot) and len(k1[0]) <= groups and len(np.unique(k1[0])) == len(k1[0]) and
                (len(k1[0]) == 0 or (k1[0].min() >= 0 and k1[0].max() < groups))):
            return False, f"grouped: {len(k1[0])} groups, count {int(np.sum(k1[2]))} sum {tot!r} against {want_cnt} {want_sum!r}"
        k3 = by_key(agg([Column(0)], [sum_v]))  # the one-aggregate form (narrow rows, lean pass 2, specialised waves)
        if not (np.array_equal(k3[0], k1[0]) and np.array_equal(k3[1].view(np.uint64), k1[1].view(np.uint64))):
            return False, "SUM alone differs from SUM beside COUNT"
        if it % 3 == 0:
            k2 = by_key(agg([Column(0)], [sum_v, count_v]))
            if not all(np.array_equal(x.view(np.uint64) if x.dtype == np.float64 else x, y.view(np.uint64) if y.dtype == np.float64 else y) for x, y in zip(k1, k2)):
                return False, "a second run of the grouped query differs from the first"
        # differential: another kernel family must give the same bits (bounded size: the global-atomic table does 24 G rows/s)
        if n <= (1 << 26):
            t3 = ex.DeviceTable.synth(syn + [("w", ex.SYNTH_F64_EXACT, 2, 0.0, 0.0)], seed, 0, n)
            shapes = [pred, BinaryExpr(Column(1), ops[1], lit(hi)), BinaryExpr(pred, Operator.And, BinaryExpr(Column(0), Operator.GtEq, Literal(ScalarValue.Int64(0)))), None,
                      BinaryExpr(BinaryExpr(Column(2), Operator.GtEq, lit(lo)), Operator.And, BinaryExpr(Column(1), Operator.Lt, lit(hi)))]
            p3 = shapes[int(rng.integers(0, len(shapes)))]
            aggs = aggregate_sets()[int(rng.integers(0, 10))]
            group = [Column(0)] if rng.random() < 0.8 else []
            key, alt, dflt = ALTERNATIVES[int(rng.integers(0, len(ALTERNATIVES)))]

            def run3():
                rel = t3.scan(batch)
                if p3 is not None:
                    rel = ex.FilterRelation(rel, ex.compile_scalar_expr(None, p3, schema3), schema3)
                rel = ex.AggregateRelation(None, rel, [ex.compile_scalar_expr(None, g, schema3) for g in group], [ex.compile_expr(None, x, schema3) for x in aggs])
                return rel.next()
            base = run3()
            ex.set_option(key, alt)
            # the first run left its strategy decision in the table's memo: every other time the alternative run takes it again from a
            # calibration slice of its own (round 6: a slice that overflows a table of 2^14 slots was a path no run had taken)
            fresh = rng.random() < 0.5
            if fresh:
                ex.set_option("agg.calibration_memo", 0)
            try:
                other = run3()
            finally:
                ex.set_option(key, dflt)
                if fresh:
                    ex.set_option("agg.calibration_memo", 1)
            same3 = same_batches(base, other) if group else (base.num_rows == other.num_rows == 1 and all(
                base.column(c)[0].as_py() == other.column(c)[0].as_py() or (base.column(c)[0].as_py() != base.column(c)[0].as_py()) for c in range(base.num_columns)))
            if not same3:
                return False, f"{key}={alt} gives another result than the defaults: predicate {p3!r}, aggregates {aggs!r}, group {bool(group)}"
        return True, ""
    try:
        ok, why = soak_case()
    except Exception as e:  # an error is a failure of the case too
        ok, why = False, f"{type(e).__name__}: {e}"
    if not ok:
        print("SOAK FAILED", case, "--", why, flush=True)
        sys.exit(1)
    del table
    if it % 20 == 0:
        print(f"  {it} iterations ok ({case})", flush=True)
print(f"soak ok: {it} iterations in {budget:.0f} s, pass-2 launches {ex.counter_get('agg_pass2_launches')}, table growths {ex.counter_get('agg_growths')}")
