"""CPU tests of the host side of the product: the C-ABI library loads and exports every symbol
include/dfx.h declares, the expression compiler mirrors the reference's names / types / errors, and
the execution path fails loudly (no CPU fallback) when there is no GPU.  No compute calls here."""
import ctypes
import os
import re

import pyarrow as pa
import pytest

from datafusion_archive_amd import _ffi
from datafusion_archive_amd import execution as ex
from datafusion_archive_amd.logicalplan import (AggregateFunction, BinaryExpr, Cast, Column, DataType, IsNull,
                                                Literal, Operator, ScalarFunction, ScalarValue, Sort, serialize)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCHEMA = pa.schema([pa.field("city", pa.string(), False), pa.field("lat", pa.float64(), False),
                    pa.field("lng", pa.float64(), False), pa.field("n", pa.int32(), False)])


def test_library_exports_every_declared_symbol():
    L = _ffi.lib()
    header = open(os.path.join(ROOT, "include", "dfx.h")).read()
    declared = sorted(set(re.findall(r"^(?:int32_t|int64_t|uint32_t|uint64_t|void|const char\*|const void\*)\s+(dfx_[a-z0-9_]+)\(", header, re.M)))
    assert declared, "no declarations parsed from include/dfx.h"
    for name in declared:
        assert hasattr(L, name), f"libdfx_hip.so does not export {name}"
    assert sorted(_ffi.EXPORTED_SYMBOLS) == declared
    assert L.dfx_abi_version() == 1


def test_expr_node_layout_matches_header():
    # include/dfx.h: 8 x int32, an 8-byte union, a pointer
    assert ctypes.sizeof(_ffi.ExprNode) == 8 * 4 + 8 + ctypes.sizeof(ctypes.c_void_p)


def test_compile_scalar_expr_names_and_types():
    """expression.rs: literal -> format!("{}", n); column -> field name; binary -> Debug of the Expr."""
    c = lambda e: ex.compile_scalar_expr(None, e, SCHEMA)
    assert c(Column(1)).get_name() == "lat" and c(Column(1)).get_type() == DataType.Float64
    assert c(Literal(ScalarValue.Float64(51.0))).get_name() == "51"
    assert c(Literal(ScalarValue.Int64(53))).get_name() == "53"
    gt = BinaryExpr(Column(1), Operator.Gt, Literal(ScalarValue.Float64(51.0)))
    lt = BinaryExpr(Column(1), Operator.Lt, Cast(Literal(ScalarValue.Int64(53)), DataType.Float64))
    e = c(BinaryExpr(gt, Operator.And, lt))
    assert e.get_name() == "#1 Gt Float64(51.0) And #1 Lt CAST(Int64(53) AS Float64)"  # sqlplanner.rs:576-584 style
    assert e.get_type() == DataType.Boolean
    assert c(BinaryExpr(Column(1), Operator.Plus, Column(2))).get_name() == "#1 Plus #2"
    assert c(BinaryExpr(Column(1), Operator.Plus, Column(2))).get_type() == DataType.Float64  # op_type = left type
    assert c(Cast(Column(1), DataType.Int32)).get_name() == "lat"  # expression.rs:323
    assert c(Cast(Literal(ScalarValue.Int64(53)), DataType.Float64)).get_name() == "lit"  # :353


@pytest.mark.parametrize("expr,kind,needle", [
    (Literal(ScalarValue.Utf8("x")), "ExecutionError", "No support for literal type"),      # :306-309
    (Literal(ScalarValue.Boolean(True)), "ExecutionError", "No support for literal type"),
    (BinaryExpr(Column(1), Operator.Modulus, Column(2)), "ExecutionError", "operator: Modulus"),  # :494-497
    (BinaryExpr(Column(1), Operator.Like, Column(2)), "ExecutionError", "operator: Like"),
    (IsNull(Column(1)), "ExecutionError", "expression #1 IS NULL"),                           # :500-503
    (Sort(Column(1), True), "ExecutionError", "expression #1 ASC"),
    (ScalarFunction("sqrt", (Column(1),), DataType.Float64), "ExecutionError", "expression sqrt(#1)"),
    (AggregateFunction("min", [Column(1)], DataType.Float64), "ExecutionError", "expression min(#1)"),
    (Cast(Column(0), DataType.Int32), "InternalError", "unsupported CAST operation"),         # :336 panic
    (Cast(Column(1), DataType.Utf8), "NotImplemented", "CAST from Float64 to Utf8"),
    (Column(9), "InternalError", "index out of bounds"),
])
def test_compile_scalar_expr_errors_mirror_reference(expr, kind, needle):
    with pytest.raises(ex.ExecutionError) as ei:
        ex.compile_scalar_expr(None, expr, SCHEMA)
    assert ei.value.kind == kind and needle in ei.value.message


def test_compile_expr_aggregates():
    """expression.rs:80-121."""
    a = ex.compile_expr(None, AggregateFunction("MIN", [Column(1)], DataType.Float64), SCHEMA)
    assert a.is_aggregate() and a.get_name() == "MIN" and a.get_type() == DataType.Float64
    assert ex.compile_expr(None, AggregateFunction("Count", [Column(0)], DataType.UInt64), SCHEMA).is_aggregate()
    # deviation D7: "avg" passes the reference's planner but not its executor (:103-106); here it compiles
    e = ex.compile_expr(None, AggregateFunction("avg", [Column(1)], DataType.Float64), SCHEMA)
    assert e.get_name() == "avg" and e.get_type() == DataType.Float64
    with pytest.raises(ex.ExecutionError) as ei:  # :103-106
        ex.compile_expr(None, AggregateFunction("median", [Column(1)], DataType.Float64), SCHEMA)
    assert ei.value.kind == "General" and "Unsupported aggregate function 'median'" in ei.value.message
    with pytest.raises(ex.ExecutionError) as ei:  # assert_eq!(1, args.len()) :91
        ex.compile_expr(None, AggregateFunction("min", [Column(1), Column(2)], DataType.Float64), SCHEMA)
    assert ei.value.kind == "InternalError"
    # a scalar expression falls through to compile_scalar_expr (:119)
    assert not ex.compile_expr(None, Column(1), SCHEMA).is_aggregate()


def test_serialization_is_children_before_parents():
    s = serialize([BinaryExpr(Column(1), Operator.Plus, Cast(Column(2), DataType.Float64))])
    assert s.roots == [3] and s.nodes[3].left == 0 and s.nodes[3].right == 2 and s.nodes[2].left == 1


def test_no_cpu_fallback_without_gpu():
    """Operators can be built without a device; pulling a batch must fail loudly, never compute on CPU."""
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    b = pa.RecordBatch.from_pydict({"city": ["a"], "lat": [52.0], "lng": [1.0], "n": pa.array([1], pa.int32())}, schema=SCHEMA)
    src = ex.DataSourceRelation(SCHEMA, [b])
    f = ex.FilterRelation(src, ex.compile_scalar_expr(None, BinaryExpr(Column(1), Operator.Gt, Column(2)), SCHEMA), SCHEMA)
    with pytest.raises(ex.ExecutionError) as ei:
        f.next()
    assert "no CPU fallback" in ei.value.message
    with pytest.raises(ex.ExecutionError):
        ex.DeviceTable.synth([("v", ex.SYNTH_F64_EXACT, 0, 0.0, 0.0)], 1, 0, 10)
    # the sources / operators added next to the path fail the same way: the CSV text is parsed on the device or not at all
    with pytest.raises(ex.ExecutionError) as ei:
        ex.CsvDataSource(os.path.join(ROOT, "tests", "data", "uk_cities.csv"), SCHEMA, 1024)
    assert "no CPU fallback" in ei.value.message
    srt = ex.SortRelation(ex.DataSourceRelation(SCHEMA, [b]), [(ex.compile_scalar_expr(None, Column(1), SCHEMA), True)], SCHEMA)
    with pytest.raises(ex.ExecutionError) as ei:
        srt.next()
    assert "no CPU fallback" in ei.value.message
    lim = ex.LimitRelation(ex.DataSourceRelation(SCHEMA, [b]), 1, SCHEMA)
    with pytest.raises(ex.ExecutionError) as ei:
        lim.next()
    assert "no CPU fallback" in ei.value.message


def test_consumed_relation_cannot_be_pulled():
    b = pa.RecordBatch.from_pydict({"city": ["a"], "lat": [52.0], "lng": [1.0], "n": pa.array([1], pa.int32())}, schema=SCHEMA)
    src = ex.DataSourceRelation(SCHEMA, [b])
    ex.ProjectRelation(src, [ex.compile_scalar_expr(None, Column(1), SCHEMA)], None)
    with pytest.raises(ex.ExecutionError):
        src.next()


def _build_c_abi_smoke(tmp_path):
    """gcc-compile the plain-C consumer of include/dfx.h against the built library (no Python, no torch)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "datafusion_archive_amd", "lib")
    exe = os.path.join(str(tmp_path), "smoke")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                    os.path.join(root, "tests", "c_abi", "smoke.c"), "-L", libdir, "-ldfx_hip",
                    "-Wl,-rpath," + libdir, "-o", exe], check=True)
    return exe


def _build_c_abi_program(tmp_path, name):
    """gcc-compile tests/c_abi/<name>.c against the built library"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "datafusion_archive_amd", "lib")
    exe = os.path.join(str(tmp_path), name)
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                    os.path.join(root, "tests", "c_abi", name + ".c"), "-L", libdir, "-ldfx_hip",
                    "-Wl,-rpath," + libdir, "-o", exe], check=True)
    return exe


def test_c_producer_of_host_batches_compiles(tmp_path):
    """tests/c_abi/host_stream.c (a plain C Arrow stream producer that poisons its buffers on release) builds against the
    header; without a GPU it fails loudly like every other entry point (its GPU run: test_gpu_parity.py)."""
    import subprocess
    import torch
    exe = _build_c_abi_program(tmp_path, "host_stream")
    if not torch.cuda.is_available():
        r = subprocess.run([exe, "1000", "2"], capture_output=True, text=True)
        assert r.returncode != 0 and r.stdout.startswith("ERR"), r.stdout + r.stderr


def test_c_abi_header_compiles_as_c_and_fails_loudly_without_gpu(tmp_path):
    """include/dfx.h is valid C11, a C program links against the library, and on a machine without a GPU the
    very first call reports an error instead of falling back to a CPU path."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    exe = _build_c_abi_smoke(tmp_path)
    r = subprocess.run([exe, "1000", "10"], capture_output=True, text=True)
    assert r.returncode != 0 and r.stdout.startswith("ERR"), r.stdout + r.stderr
    assert "no CPU" in r.stdout or "HIP" in r.stdout, r.stdout


def test_rust_binding_matches_header():
    """integration/rust/src/execution/gpu.rs cannot be compiled here (no Rust toolchain): at least keep its extern "C" block
    in step with include/dfx.h -- every bound function exists in the header with the same number of parameters."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "dfx.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    rust = open(os.path.join(root, "integration", "rust", "src", "execution", "gpu.rs")).read()
    block = rust[rust.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    # the link directive decorates the extern block itself (on anything else the symbols would stay undefined at link time)
    assert re.search(r'#\[link\(name = "dfx_hip"\)\]\s*\n\s*extern "C" \{', rust), "#[link] must sit directly on the extern block"

    def n_params(arglist):
        arglist = arglist.strip()
        return 0 if arglist in ("", "void") else arglist.count(",") + 1

    c_protos = {m.group(1): n_params(m.group(2)) for m in re.finditer(r"\b(dfx_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", header)}
    rs_protos = {m.group(1): n_params(m.group(2)) for m in re.finditer(r"fn (dfx_[a-z0-9_]+)\s*\(([^()]*)\)", block)}
    assert len(rs_protos) >= 18
    for name, n in rs_protos.items():
        assert name in c_protos, f"{name} is bound in gpu.rs but not declared in include/dfx.h"
        assert c_protos[name] == n, f"{name}: {n} parameters in gpu.rs, {c_protos[name]} in include/dfx.h"
    # the operators, the data sources and the resident table are all reachable from Rust
    for must in ("dfx_filter_relation_new", "dfx_project_relation_new", "dfx_aggregate_relation_new", "dfx_csv_datasource_new",
                 "dfx_sort_relation_new", "dfx_limit_relation_new", "dfx_table_from_stream", "dfx_table_scan_new",
                 # the multi-GPU path: the three device steps and the in-library RCCL exchange
                 "dfx_aggregate_partial_build", "dfx_aggregate_partial_export", "dfx_aggregate_partial_import",
                 "dfx_comm_unique_id", "dfx_comm_init", "dfx_comm_destroy", "dfx_aggregate_exchange"):
        assert must in rs_protos


def test_c_abi_rejects_null_arguments():
    """Every constructor of include/dfx.h answers a null / missing argument with a status code and a message: nothing is
    dereferenced blindly and nothing unwinds across the boundary (needs no GPU: the checks precede any device work)."""
    import ctypes
    from datafusion_archive_amd import _ffi
    L = _ffi.lib()
    err = ctypes.create_string_buffer(512)
    out = _ffi.ArrowArrayStream()
    handle = ctypes.c_void_p()
    N = None
    calls = {
        "filter": lambda: L.dfx_filter_relation_new(N, N, N, ctypes.byref(out), err, 512),
        "filter without out": lambda: L.dfx_filter_relation_new(N, N, N, N, err, 512),
        "project": lambda: L.dfx_project_relation_new(N, N, 0, N, ctypes.byref(out), err, 512),
        "aggregate": lambda: L.dfx_aggregate_relation_new(N, N, N, 0, N, 0, ctypes.byref(out), err, 512),
        "sort": lambda: L.dfx_sort_relation_new(N, N, N, 0, N, ctypes.byref(out), err, 512),
        "limit": lambda: L.dfx_limit_relation_new(N, 5, N, ctypes.byref(out), err, 512),
        "csv": lambda: L.dfx_csv_datasource_new(N, N, 10, ctypes.byref(out), err, 512),
        "compile_scalar_expr": lambda: L.dfx_compile_scalar_expr(N, 0, 0, N, ctypes.byref(handle), err, 512),
        "compile_expr": lambda: L.dfx_compile_expr(N, 0, 0, N, ctypes.byref(handle), err, 512),
        "table_scan": lambda: L.dfx_table_scan_new(N, 10, ctypes.byref(out), err, 512),
    }
    for name, call in calls.items():
        err.value = b""
        rc = call()
        assert rc != 0, name
        assert err.value, f"{name}: status {rc} without a message"
    # a caller that passes no error buffer still gets the code
    assert L.dfx_filter_relation_new(N, N, N, ctypes.byref(out), N, 0) != 0


# ---- EXPLAIN: which kernel family every BASELINE query shape will run on (decided on the host, no GPU needed) ------------
def _explain_aggregate(schema, filter_expr, group, aggs):
    batch = pa.RecordBatch.from_pydict({f.name: pa.array([1], f.type) for f in schema}, schema=schema)
    rel = ex.DataSourceRelation(schema, [batch])
    if filter_expr is not None:
        rel = ex.FilterRelation(rel, ex.compile_scalar_expr(None, filter_expr, schema), schema)
    rel = ex.AggregateRelation(None, rel, [ex.compile_scalar_expr(None, g, schema) for g in group],
                               [ex.compile_expr(None, a, schema) for a in aggs])
    return ex.explain(rel)


def test_explain_baseline_shapes_hit_their_static_signatures():
    """A query that silently misses its compile-time signature still gives the right answer, only slower: pin the
    plan of every BASELINE.json shape (bench.py builds exactly these trees)."""
    f64 = DataType.Float64
    kv = pa.schema([("k", pa.int64()), ("v", pa.float64())])

    def l64(v):
        return Literal(ScalarValue.Float64(v))
    pred = BinaryExpr(BinaryExpr(Column(1), Operator.Gt, l64(204.8)), Operator.And, BinaryExpr(Column(1), Operator.Lt, l64(409.6)))
    sum_v = AggregateFunction("SUM", [Column(1)], f64)
    count_v = AggregateFunction("COUNT", [Column(1)], DataType.UInt64)
    # the headline query: filter + GROUP BY SUM; the Filter is absorbed by the aggregate
    text = _explain_aggregate(kv, pred, [Column(0)], [sum_v])
    assert "static shape KeySumPred2F64" in text and "Filter below fused" in text and "\n  HostStream" in text
    assert "Filter:" not in text  # no separate operator left
    # config 3: no predicate
    assert "static shape KeySum," in _explain_aggregate(kv, None, [Column(0)], [sum_v])
    # config 2 flavours: predicate + COUNT, predicate + SUM + COUNT (bench.py's verification query)
    assert "static shape CountPred2F64" in _explain_aggregate(kv, pred, [], [count_v])
    assert "static shape SumCountPred2F64" in _explain_aggregate(kv, pred, [], [sum_v, count_v])
    # config 5: the TPC-H Q1 shape
    names = ["rf", "ls", "qty", "price", "disc", "tax", "ship"]
    q1 = pa.schema([(n, pa.int64() if i < 2 else pa.float64()) for i, n in enumerate(names)])
    dp = BinaryExpr(Column(3), Operator.Multiply, BinaryExpr(l64(1.0), Operator.Minus, Column(4)))
    aggs = [AggregateFunction("sum", [Column(2)], f64), AggregateFunction("sum", [Column(3)], f64), AggregateFunction("sum", [dp], f64),
            AggregateFunction("sum", [BinaryExpr(dp, Operator.Multiply, BinaryExpr(l64(1.0), Operator.Plus, Column(5)))], f64)]
    pred5 = BinaryExpr(BinaryExpr(Column(6), Operator.LtEq, l64(2436.0)), Operator.And, BinaryExpr(Column(4), Operator.GtEq, l64(0.0)))
    text = _explain_aggregate(q1, pred5, [Column(0), Column(1)], aggs)
    assert "static shape Q1" in text and "2 keys, 4 accumulators" in text and "7 columns" in text
    # one ordered Float64 comparison = a two-sided range with an infinite bound (round 3): the same signatures
    for op in (Operator.Lt, Operator.LtEq, Operator.Gt, Operator.GtEq):
        one = BinaryExpr(Column(1), op, l64(204.8))
        assert "static shape KeySumPred2F64" in _explain_aggregate(kv, one, [Column(0)], [sum_v]), op
        assert "static shape KeySumPred2F64" in _explain_aggregate(kv, BinaryExpr(l64(204.8), op, Column(1)), [Column(0)], [sum_v]), op
        assert "static shape CountPred2F64" in _explain_aggregate(kv, one, [], [count_v]), op
    # round 4: what the signatures do not name is a SCAN PLAN (the query as data: range tests on value images), not a decoded shape
    assert "scan plan (PlanPolicy" in _explain_aggregate(kv, BinaryExpr(Column(1), Operator.Eq, l64(204.8)), [Column(0)], [sum_v])
    kvw = pa.schema([("k", pa.int64()), ("v", pa.float64()), ("w", pa.float64())])  # three columns: no signature to win, the term stays one term
    assert "scan plan (PlanPolicy" in _explain_aggregate(kvw, BinaryExpr(Column(1), Operator.Lt, l64(204.8)), [Column(0)], [AggregateFunction("SUM", [Column(2)], f64)])
    assert "scan plan (PlanPolicy" in _explain_aggregate(kv, BinaryExpr(Column(0), Operator.Lt, Literal(ScalarValue.Int64(5))), [Column(0)], [sum_v])
    # ... any single MIN / MAX / COUNT / SUM, an Int32 key, a Float32 / Int32 predicate column, a third term, nulls in any of them
    k32 = pa.schema([("k", pa.int32()), ("v", pa.float64()), ("w", pa.float32())])
    three = BinaryExpr(pred, Operator.And, BinaryExpr(Column(0), Operator.GtEq, Literal(ScalarValue.Int32(0))))
    for agg in (AggregateFunction("MIN", [Column(1)], f64), AggregateFunction("MAX", [Column(1)], f64), AggregateFunction("COUNT", [Column(1)], DataType.UInt64), sum_v):
        text = _explain_aggregate(k32, three, [Column(0)], [agg])
        assert "scan plan (PlanPolicy" in text and "batches with nulls too" in text, text
    wpred = BinaryExpr(Column(2), Operator.Lt, Literal(ScalarValue.Float32(0.5)))
    assert "scan plan (PlanPolicy" in _explain_aggregate(k32, wpred, [Column(0)], [sum_v])
    # not a plan: a Float32 argument of SUM (the accumulators take the argument's own bits), a product argument, an Int16 column
    assert "scan plan" not in _explain_aggregate(k32, wpred, [Column(0)], [AggregateFunction("SUM", [Column(2)], DataType.Float32)])
    k16 = pa.schema([("k", pa.int16()), ("v", pa.float64())])
    assert "scan plan" not in _explain_aggregate(k16, None, [Column(0)], [sum_v])
    # SUM(column <op> literal) under the headline's predicate: its own pass-1 signature (round 3)
    for arg in (BinaryExpr(Column(1), Operator.Multiply, l64(2.0)), BinaryExpr(l64(2.0), Operator.Multiply, Column(1)),
                BinaryExpr(Column(1), Operator.Plus, l64(2.0)), BinaryExpr(Column(1), Operator.Minus, l64(2.0)), BinaryExpr(l64(2.0), Operator.Minus, Column(1))):
        assert "static shape KeyAffSumPred2F64" in _explain_aggregate(kv, pred, [Column(0)], [AggregateFunction("SUM", [arg], f64)])
    # anything else: the run-time decoded shape family, or the interpreter
    assert "scan plan (PlanPolicy" in _explain_aggregate(kv, pred, [Column(0)], [AggregateFunction("MAX", [Column(1)], f64)])
    prod = BinaryExpr(BinaryExpr(Column(1), Operator.Multiply, l64(2.0)), Operator.Multiply, BinaryExpr(Column(1), Operator.Plus, l64(1.0)))
    assert "FastPolicy" in _explain_aggregate(kv, BinaryExpr(Column(1), Operator.Eq, l64(204.8)), [Column(0)], [AggregateFunction("SUM", [prod], f64)])
    # AVG = SUM + COUNT of one operand, SUM + MIN + MAX of one column: shared routed value; different operands: not
    assert "2 aggregates of ONE operand" in _explain_aggregate(kv, pred, [Column(0)], [AggregateFunction("AVG", [Column(1)], f64)])
    assert "3 aggregates of ONE operand" in _explain_aggregate(kv, None, [Column(0)], [sum_v, AggregateFunction("MIN", [Column(1)], f64), AggregateFunction("MAX", [Column(1)], f64)])
    assert "ONE operand" not in _explain_aggregate(kv, None, [Column(0)], [sum_v, AggregateFunction("MAX", [Column(0)], DataType.Int64)])
    assert "ONE operand" not in _explain_aggregate(kv, pred, [Column(0)], [sum_v])
    # ... aggregates of DIFFERENT operands under one key: one scan per aggregate if the partitioned strategy is chosen (round 4)
    text = _explain_aggregate(kvw, pred, [Column(0)], [sum_v, AggregateFunction("MIN", [Column(2)], f64)])
    assert "2 aggregates of different operands" in text and "one scan per aggregate" in text, text
    assert "different operands" not in _explain_aggregate(kv, pred, [Column(0)], [sum_v])
    assert "different operands" not in _explain_aggregate(kv, pred, [Column(0)], [AggregateFunction("AVG", [Column(1)], f64)])
    assert "different operands" not in _explain_aggregate(kvw, pred, [Column(0), Column(2)], [sum_v, AggregateFunction("MIN", [Column(2)], f64)])  # two keys
    assert "SSA interpreter" in _explain_aggregate(kv, pred, [Column(0)], [AggregateFunction("SUM", [BinaryExpr(Column(1), Operator.Plus, Column(1))], f64)])


def test_per_operator_options_are_validated_and_do_not_touch_the_process_defaults():
    """include/dfx.h: dfx_*_relation_new_with_options.  An operator takes its own option set (process defaults + its
    overrides); an unknown key is General; nothing global changes (no GPU needed: the constructors do no device work)."""
    schema = pa.schema([("k", pa.int64()), ("v", pa.float64())])
    b = pa.RecordBatch.from_pydict({"k": [1], "v": [2.0]}, schema=schema)
    pred = BinaryExpr(Column(1), Operator.Gt, Literal(ScalarValue.Float64(1.0)))
    sum_v = AggregateFunction("SUM", [Column(1)], DataType.Float64)

    def tree(filter_opts=None, agg_opts=None):
        rel = ex.FilterRelation(ex.DataSourceRelation(schema, [b]), ex.compile_scalar_expr(None, pred, schema), schema, options=filter_opts)
        return ex.AggregateRelation(None, rel, [ex.compile_scalar_expr(None, Column(0), schema)], [ex.compile_expr(None, sum_v, schema)], options=agg_opts)
    tree({"filter.single_pass": 0, "scan.fast": 0}, {"agg.strategy": 1, "agg.pass1_ws": 0})  # accepted
    with pytest.raises(ex.ExecutionError) as e:
        tree(agg_opts={"agg.no_such_knob": 1})
    assert e.value.kind == "General" and "agg.no_such_knob" in e.value.message
    with pytest.raises(ex.ExecutionError) as e:
        tree(filter_opts={"bogus": 1})
    assert e.value.kind == "General"
    with pytest.raises(ex.ExecutionError):
        ex.set_option("agg.no_such_knob", 1)  # the process-wide setter knows the same keys


def test_explain_operator_tree_and_pushdown():
    schema = pa.schema([("k", pa.int64()), ("v", pa.float64()), ("w", pa.float64())])
    b = pa.RecordBatch.from_pydict({"k": [1], "v": [2.0], "w": [3.0]}, schema=schema)
    pred = BinaryExpr(Column(1), Operator.Gt, Literal(ScalarValue.Float64(1.0)))
    f = ex.FilterRelation(ex.DataSourceRelation(schema, [b]), ex.compile_scalar_expr(None, pred, schema), schema)
    p = ex.ProjectRelation(f, [ex.compile_scalar_expr(None, Column(0), schema),
                               ex.compile_scalar_expr(None, BinaryExpr(Column(1), Operator.Multiply, Column(1)), schema)], None)
    out_schema = pa.schema([("k", pa.int64()), ("vv", pa.float64())])
    top = ex.LimitRelation(ex.SortRelation(p, [(ex.compile_scalar_expr(None, Column(1), out_schema), False)], out_schema), 5, out_schema)
    lines = ex.explain(top).splitlines()
    assert [ln.strip().split(":")[0] for ln in lines] == ["Limit", "Sort", "Project", "Project", "Filter", "HostStream"]
    assert [len(ln) - len(ln.lstrip()) for ln in lines] == [0, 2, 4, 6, 8, 10]
    assert "top-5 by radix select" in lines[1] and "DESC" in lines[1]
    assert "1 zero-copy columns, 1 fused programs" in lines[3]
    # projection push-down is visible: the projection reads k and v, so w is neither compacted nor uploaded
    assert "2 columns compacted" in lines[4]
    assert "2 of 3 columns uploaded" in lines[5]
    # a foreign stream is not explained
    import ctypes
    from datafusion_archive_amd import _ffi
    foreign = _ffi.ArrowArrayStream()
    assert _ffi.lib().dfx_relation_explain(ctypes.addressof(foreign), None, 0) == -1


def test_conjunction_beyond_one_fused_program_is_split_on_the_host():
    """FilterRelation packs the top-level AND chain of an oversized predicate into several fused programs (8 columns, 16
    computed values, 16 literals each) -- decided at construction, visible in the plan; under an aggregate such a Filter
    stays a relation of its own (no fusion).  A disjunction of the same size is not split: the limit is reported on
    next() (the reference's errors surface there as well), so construction still succeeds."""
    n_cols = 12
    schema = pa.schema([(f"c{i}", pa.float64()) for i in range(n_cols)])
    b = pa.RecordBatch.from_pydict({f"c{i}": [1.0] for i in range(n_cols)}, schema=schema)
    terms = []
    for c in range(n_cols):
        terms.append(BinaryExpr(Column(c), Operator.Gt, Literal(ScalarValue.Float64(float(c)))))
        terms.append(BinaryExpr(Column(c), Operator.Lt, Literal(ScalarValue.Float64(100.0 + c))))

    def chain(op):
        e = terms[0]
        for t in terms[1:]:
            e = BinaryExpr(e, op, t)
        return e
    f = ex.FilterRelation(ex.DataSourceRelation(schema, [b]), ex.compile_scalar_expr(None, chain(Operator.And), schema), schema)
    text = ex.explain(f)
    assert "conjunction evaluated by" in text and "fused programs (masks ANDed)" in text, text
    n_prog = int(text.split("conjunction evaluated by ")[1].split()[0])
    assert 3 <= n_prog <= 6, text  # 24 literals / 16 per program, 47 values / 16: at least three
    agg = ex.AggregateRelation(None, ex.FilterRelation(ex.DataSourceRelation(schema, [b]), ex.compile_scalar_expr(None, chain(Operator.And), schema), schema),
                               [], [ex.compile_expr(None, AggregateFunction("SUM", [Column(0)], DataType.Float64), schema)])
    lines = [ln.strip().split(":")[0] for ln in ex.explain(agg).splitlines()]
    assert lines[:2] == ["Aggregate", "Filter"], lines  # the Filter was not absorbed into the aggregate's scan
    g = ex.FilterRelation(ex.DataSourceRelation(schema, [b]), ex.compile_scalar_expr(None, chain(Operator.Or), schema), schema)
    assert "error deferred to next()" in ex.explain(g) and "more than" in ex.explain(g)


def test_group_hash_is_a_bijection_of_narrow_keys():
    """Narrow rows (PTF_NARROW) stand on this: for keys below 2^32 the 32-bit image of the group hash identifies the key
    (dfx_debug_unhash32 inverts it), images of different keys differ, and a key with high bits hashes differently from its
    low word alone (so it must not be sent as an image)."""
    import numpy as np
    from datafusion_archive_amd import _ffi
    L = _ffi.lib()
    rng = np.random.default_rng(3)
    keys = np.unique(np.concatenate([np.arange(0, 5000), rng.integers(0, 1 << 32, 200000), [(1 << 32) - 1, 999999, 1 << 31]]).astype(np.uint64))
    images = np.array([L.dfx_debug_group_hash(int(k)) >> 32 for k in keys], dtype=np.uint64)
    assert all(L.dfx_debug_group_hash(int(k)) & 0xFFFFFFFF == 0 for k in keys[:100])  # the hash lives in the high half
    assert len(np.unique(images)) == len(keys)
    for k, im in zip(keys[::37], images[::37]):
        assert L.dfx_debug_unhash32(int(im)) == int(k)
    wide = int(keys[1234]) | (5 << 32)
    assert L.dfx_debug_group_hash(wide) != L.dfx_debug_group_hash(int(keys[1234]))
