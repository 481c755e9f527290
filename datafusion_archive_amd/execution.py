"""Host-side mirror of the reference's operator interface for the hot path, over the C ABI.

Same names, argument order and error behaviour as ``src/execution`` of the reference:

* ``compile_scalar_expr`` / ``compile_expr``  (expression.rs:283 / :80) -> ``RuntimeExpr``
* ``Relation`` with ``next()`` / ``schema()``   (relation.rs:27-32)
* ``FilterRelation(input, expr, schema)``        (filter.rs:36)
* ``ProjectRelation(input, exprs, schema)``      (projection.rs:36)
* ``AggregateRelation(schema, input, group_expr, aggr_expr)`` (aggregate.rs:47-52)
* ``ExecutionError`` with the variant names of error.rs:26-36

Batches are ``pyarrow.RecordBatch`` (the Arrow C Data Interface carries them across the ABI).
This module is plumbing: all computing happens in ``libdfx_hip.so`` on the GPU.
"""
from __future__ import annotations

import ctypes
import os
from typing import Iterable, List, Optional, Sequence

import pyarrow as pa

from . import _ffi
from .logicalplan import DataType, Expr, serialize

_ERR_NAMES = {1: "IoError", 2: "ParserError", 3: "General", 4: "InvalidColumn", 5: "NotImplemented",
              6: "InternalError", 7: "ArrowError", 8: "ExecutionError"}


class ExecutionError(Exception):
    """execution::error::ExecutionError (error.rs:26-36); ``kind`` is the variant name."""

    def __init__(self, code: int, message: str):
        self.code = int(code)
        self.kind = _ERR_NAMES.get(int(code), f"Unknown({code})")
        self.message = message
        super().__init__(f"{self.kind}({message!r})")


def _check(code: int, err: ctypes.Array) -> None:
    if code != 0:
        raise ExecutionError(code, err.value.decode(errors="replace"))


def _errbuf():
    return ctypes.create_string_buffer(1024)


def _export_schema(schema: Optional[pa.Schema]):
    """pyarrow schema -> a live ArrowSchema struct (caller keeps it until the call returns)."""
    c = _ffi.ArrowSchema()
    if schema is not None:
        schema._export_to_c(ctypes.addressof(c))
    return c


def _release_schema(c: "_ffi.ArrowSchema") -> None:
    if c.release:
        ctypes.CFUNCTYPE(None, ctypes.POINTER(_ffi.ArrowSchema))(c.release)(ctypes.byref(c))


def _options(options: Optional[dict]):
    """dict -> (dfx_option array or None, count, objects to keep alive during the call)"""
    if not options:
        return None, 0, None
    keys = [k.encode() for k in options]
    arr = (_ffi.OptionC * len(keys))()
    for i, (k, v) in enumerate(zip(keys, options.values())):
        arr[i].key, arr[i].value = k, int(v)
    return arr, len(keys), keys


class RuntimeExpr:
    """expression::RuntimeExpr (expression.rs:42-77)."""

    def __init__(self, handle: int, expr: Expr):
        self._h = ctypes.c_void_p(handle)
        self.expr = expr

    def get_name(self) -> str:
        return _ffi.lib().dfx_runtime_expr_name(self._h).decode()

    def get_type(self) -> DataType:
        return DataType(_ffi.lib().dfx_runtime_expr_type(self._h))

    def is_aggregate(self) -> bool:
        return bool(_ffi.lib().dfx_runtime_expr_is_aggregate(self._h))

    def __del__(self):
        try:
            if self._h:
                _ffi.lib().dfx_runtime_expr_free(self._h)
                self._h = None
        except Exception:
            pass


def _compile(fn_name: str, expr: Expr, input_schema: pa.Schema) -> RuntimeExpr:
    L = _ffi.lib()
    s = serialize([expr])
    cs = _export_schema(input_schema)
    out = ctypes.c_void_p()
    err = _errbuf()
    try:
        code = getattr(L, fn_name)(s.nodes, s.n_nodes, s.roots[0], ctypes.byref(cs), ctypes.byref(out), err, 1024)
    finally:
        _release_schema(cs)
    _check(code, err)
    return RuntimeExpr(out.value, expr)


def compile_scalar_expr(ctx, expr: Expr, input_schema: pa.Schema) -> RuntimeExpr:
    """expression::compile_scalar_expr(ctx, expr, input_schema) (expression.rs:283). ``ctx`` is unused
    (the reference only threads it through)."""
    return _compile("dfx_compile_scalar_expr", expr, input_schema)


def compile_expr(ctx, expr: Expr, input_schema: pa.Schema) -> RuntimeExpr:
    """expression::compile_expr(ctx, expr, input_schema) (expression.rs:80): also accepts aggregates."""
    return _compile("dfx_compile_expr", expr, input_schema)


class Relation:
    """trait Relation (relation.rs:27-32) over an ArrowArrayStream owned by this object."""

    def __init__(self):
        self._stream = _ffi.ArrowArrayStream()
        self._schema: Optional[pa.Schema] = None
        self._keep: list = []

    # -- protocol ---------------------------------------------------------------------------------
    def next(self) -> Optional[pa.RecordBatch]:
        """Ok(Some(batch)) -> RecordBatch, Ok(None) -> None, Err(e) -> raises ExecutionError."""
        st = self._live_stream()
        arr = _ffi.ArrowArray()
        code = st.get_next(ctypes.byref(st), ctypes.byref(arr))
        if code != 0:
            msg = st.get_last_error(ctypes.byref(st))
            raise ExecutionError(code, (msg or b"").decode(errors="replace"))
        if not arr.release:
            return None
        cs = _ffi.ArrowSchema()
        code = st.get_schema(ctypes.byref(st), ctypes.byref(cs))
        if code != 0:
            raise ExecutionError(code, "get_schema failed")
        return pa.RecordBatch._import_from_c(ctypes.addressof(arr), ctypes.addressof(cs))

    def schema(self) -> pa.Schema:
        if self._schema is None:
            st = self._live_stream()
            cs = _ffi.ArrowSchema()
            code = st.get_schema(ctypes.byref(st), ctypes.byref(cs))
            if code != 0:
                raise ExecutionError(code, "get_schema failed")
            self._schema = pa.Schema._import_from_c(ctypes.addressof(cs))
        return self._schema

    def __iter__(self):
        while True:
            b = self.next()
            if b is None:
                return
            yield b

    # -- plumbing ---------------------------------------------------------------------------------
    def _live_stream(self) -> "_ffi.ArrowArrayStream":
        if not self._stream.release:
            raise ExecutionError(3, "relation has been consumed by another operator (or closed)")
        return self._stream

    def _take_stream(self) -> "_ffi.ArrowArrayStream":
        """Hands the C stream to a consuming operator (moves it)."""
        return self._live_stream()

    def close(self) -> None:
        if self._stream.release:
            self._stream.release(ctypes.byref(self._stream))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DataSourceRelation(Relation):
    """relation::DataSourceRelation (relation.rs:34-54) over host Arrow batches (the DataSource)."""

    def __init__(self, schema: pa.Schema, batches: Iterable[pa.RecordBatch]):
        super().__init__()
        reader = pa.RecordBatchReader.from_batches(schema, iter(batches))
        reader._export_to_c(ctypes.addressof(self._stream))
        self._schema = schema


class CsvDataSource(Relation):
    """datasource::CsvDataSource::new(filename, schema, batch_size) (datasource.rs:39-43) wrapped in its
    DataSourceRelation (relation.rs:34-54).  The first record is always consumed as a header, as in the reference.
    Parsing runs on the device; stacked operators consume the batches without a host round trip."""

    def __init__(self, filename: str, schema: pa.Schema, batch_size: int = 1024):
        super().__init__()
        cs = _export_schema(schema)
        err = _errbuf()
        try:
            code = _ffi.lib().dfx_csv_datasource_new(os.fsencode(filename), ctypes.byref(cs), batch_size,
                                                     ctypes.byref(self._stream), err, 1024)
        finally:
            _release_schema(cs)
        _check(code, err)
        self._schema = schema


class FilterRelation(Relation):
    """filter::FilterRelation::new(input, expr, schema) (filter.rs:36)."""

    def __init__(self, input: Relation, expr: RuntimeExpr, schema: Optional[pa.Schema] = None, options: Optional[dict] = None):
        """`options`: per-operator option set (include/dfx.h: dfx_option), e.g. {"filter.single_pass": 0}; not in the reference."""
        super().__init__()
        L = _ffi.lib()
        cs = _export_schema(schema)
        err = _errbuf()
        opts, n_opts, _keep = _options(options)
        try:
            code = L.dfx_filter_relation_new_with_options(ctypes.byref(input._take_stream()), expr._h, ctypes.byref(cs), opts, n_opts,
                                                          ctypes.byref(self._stream), err, 1024)
        finally:
            _release_schema(cs)
        _check(code, err)
        self._keep = [input, expr]


    # test hook (include/dfx.h: dfx_filter_debug_mask)
    def keep_mask(self) -> None:
        err = _errbuf()
        _check(_ffi.lib().dfx_filter_debug_mask(ctypes.byref(self._live_stream()), None, 0, None, err, 1024), err)

    def last_mask(self, rows: int):
        """The LSB-first bitmap of the most recent input batch (numpy uint8, (rows + 7) // 8 bytes) and its row count."""
        import numpy as np
        buf = np.zeros((rows + 7) // 8, dtype=np.uint8)
        got = ctypes.c_int64()
        err = _errbuf()
        _check(_ffi.lib().dfx_filter_debug_mask(ctypes.byref(self._live_stream()), buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes,
                                                ctypes.byref(got), err, 1024), err)
        return buf[:(got.value + 7) // 8], got.value


class ProjectRelation(Relation):
    """projection::ProjectRelation::new(input, expr, schema) (projection.rs:36)."""

    def __init__(self, input: Relation, expr: Sequence[RuntimeExpr], schema: Optional[pa.Schema] = None):
        super().__init__()
        L = _ffi.lib()
        cs = _export_schema(schema)
        hs = (ctypes.c_void_p * max(1, len(expr)))(*[e._h for e in expr])
        err = _errbuf()
        try:
            code = L.dfx_project_relation_new(ctypes.byref(input._take_stream()), hs, len(expr), ctypes.byref(cs),
                                              ctypes.byref(self._stream), err, 1024)
        finally:
            _release_schema(cs)
        _check(code, err)
        self._keep = [input, list(expr)]


class SortRelation(Relation):
    """The operator behind LogicalPlan::Sort (logicalplan.rs:327-332; the reference's executor has none, context.rs:113).
    `sort_expr`: [(RuntimeExpr of the inner expression of Expr::Sort, asc)]."""

    def __init__(self, input: Relation, sort_expr: Sequence[tuple], schema: Optional[pa.Schema] = None):
        super().__init__()
        cs = _export_schema(schema)
        hs = (ctypes.c_void_p * max(1, len(sort_expr)))(*[e._h for e, _ in sort_expr])
        asc = (ctypes.c_int32 * max(1, len(sort_expr)))(*[1 if a else 0 for _, a in sort_expr])
        err = _errbuf()
        try:
            code = _ffi.lib().dfx_sort_relation_new(ctypes.byref(input._take_stream()), hs, asc, len(sort_expr),
                                                    ctypes.byref(cs), ctypes.byref(self._stream), err, 1024)
        finally:
            _release_schema(cs)
        _check(code, err)
        self._keep = [input] + [e for e, _ in sort_expr]


class LimitRelation(Relation):
    """The operator behind LogicalPlan::Limit { limit, input, schema } (logicalplan.rs:313-318)."""

    def __init__(self, input: Relation, limit: int, schema: Optional[pa.Schema] = None):
        super().__init__()
        cs = _export_schema(schema)
        err = _errbuf()
        try:
            code = _ffi.lib().dfx_limit_relation_new(ctypes.byref(input._take_stream()), limit, ctypes.byref(cs),
                                                     ctypes.byref(self._stream), err, 1024)
        finally:
            _release_schema(cs)
        _check(code, err)
        self._keep = [input]


class AggregateRelation(Relation):
    """aggregate::AggregateRelation::new(schema, input, group_expr, aggr_expr) (aggregate.rs:47-52)."""

    def __init__(self, schema: Optional[pa.Schema], input: Relation, group_expr: Sequence[RuntimeExpr],
                 aggr_expr: Sequence[RuntimeExpr], options: Optional[dict] = None):
        """`options`: per-operator option set (include/dfx.h: dfx_option), e.g. {"agg.strategy": 1}; not in the reference."""
        super().__init__()
        L = _ffi.lib()
        cs = _export_schema(schema)
        gs = (ctypes.c_void_p * max(1, len(group_expr)))(*[e._h for e in group_expr])
        as_ = (ctypes.c_void_p * max(1, len(aggr_expr)))(*[e._h for e in aggr_expr])
        err = _errbuf()
        opts, n_opts, _keep = _options(options)
        try:
            code = L.dfx_aggregate_relation_new_with_options(ctypes.byref(cs), ctypes.byref(input._take_stream()), gs,
                                                             len(group_expr), as_, len(aggr_expr), opts, n_opts,
                                                             ctypes.byref(self._stream), err, 1024)
        finally:
            _release_schema(cs)
        _check(code, err)
        self._keep = [input, list(group_expr), list(aggr_expr)]

    # multi-GPU GROUP BY exchange (include/dfx.h: dfx_aggregate_partial_*)
    def partial_build(self, world: int):
        """Drain the input into the local partial table. Returns (n_words, counts per destination rank)."""
        L = _ffi.lib()
        counts = (ctypes.c_int64 * world)()
        nw = ctypes.c_int32()
        err = _errbuf()
        _check(L.dfx_aggregate_partial_build(ctypes.byref(self._live_stream()), world, ctypes.byref(nw), counts,
                                             err, 1024), err)
        return nw.value, list(counts)

    def partial_export(self, dst_device_ptr: int, dst_words: int) -> None:
        L = _ffi.lib()
        err = _errbuf()
        _check(L.dfx_aggregate_partial_export(ctypes.byref(self._live_stream()), ctypes.c_void_p(dst_device_ptr),
                                              dst_words, err, 1024), err)

    def partial_import(self, src_device_ptr: int, counts: Sequence[int]) -> None:
        L = _ffi.lib()
        c = (ctypes.c_int64 * max(1, len(counts)))(*counts)
        err = _errbuf()
        _check(L.dfx_aggregate_partial_import(ctypes.byref(self._live_stream()), ctypes.c_void_p(src_device_ptr), c,
                                              len(counts), err, 1024), err)


class Communicator:
    """RCCL communicator owned by the library (include/dfx.h: dfx_comm_*), one per process/GPU.  `unique_id()` on rank 0,
    hand the 128 bytes to every rank (any host channel), then `Communicator(id, world, rank)` everywhere."""

    COMM_ID_BYTES = 128

    @staticmethod
    def unique_id() -> bytes:
        L = _ffi.lib()
        buf = ctypes.create_string_buffer(Communicator.COMM_ID_BYTES)
        err = _errbuf()
        _check(L.dfx_comm_unique_id(buf, err, 1024), err)
        return buf.raw

    def __init__(self, unique_id: bytes, world: int, rank: int):
        L = _ffi.lib()
        assert len(unique_id) == Communicator.COMM_ID_BYTES
        self._h = ctypes.c_void_p()
        self.world, self.rank = world, rank
        err = _errbuf()
        _check(L.dfx_comm_init(ctypes.create_string_buffer(unique_id, Communicator.COMM_ID_BYTES), world, rank,
                               ctypes.byref(self._h), err, 1024), err)

    def exchange(self, agg: "AggregateRelation") -> dict:
        """dfx_aggregate_exchange: counts + buckets over RCCL inside the library; then agg.next() emits the owned groups."""
        L = _ffi.lib()
        stats = (ctypes.c_int64 * 4)()
        err = _errbuf()
        _check(L.dfx_aggregate_exchange(ctypes.byref(agg._live_stream()), self._h, stats, err, 1024), err)
        return {"sent_groups": stats[0], "received_groups": stats[1], "sent_bytes": stats[2], "host_syncs": stats[3]}

    def ranks(self) -> int:
        """ncclCommCount of the library's communicator: the ranks RCCL itself sees"""
        return int(_ffi.lib().dfx_comm_ranks(self._h))

    def close(self) -> None:
        if getattr(self, "_h", None) and self._h.value:
            _ffi.lib().dfx_comm_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------------
# HBM-resident tables (the in-memory DataSource)
# ---------------------------------------------------------------------------------------------------
SYNTH_F64_UNIFORM, SYNTH_F64_EXACT, SYNTH_I64_UNIFORM, SYNTH_I64_ZIPF, SYNTH_I32_UNIFORM, SYNTH_I64_WIDE = 0, 1, 2, 3, 4, 5


def synth_nulls(kind: int, permille: int) -> int:
    """include/dfx.h: `kind | (permille << 8)` -- the column gets a validity bitmap, a row is NULL with probability permille / 1000"""
    return kind | (int(permille) << 8)


class _TableScan(Relation):
    def __init__(self, table: "DeviceTable", batch_rows: int, row_begin: int = 0, n_rows: int = -1):
        super().__init__()
        err = _errbuf()
        if row_begin == 0 and n_rows < 0:
            _check(_ffi.lib().dfx_table_scan_new(table._h, batch_rows, ctypes.byref(self._stream), err, 1024), err)
        else:
            _check(_ffi.lib().dfx_table_scan_range_new(table._h, row_begin, n_rows, batch_rows, ctypes.byref(self._stream), err, 1024), err)
        self._keep = [table]


class DeviceTable:
    """A table resident in HBM; ``scan()`` is the DataSourceRelation over it."""

    def __init__(self, handle: int):
        self._h = ctypes.c_void_p(handle)

    @staticmethod
    def from_batches(schema: pa.Schema, batches: Iterable[pa.RecordBatch]) -> "DeviceTable":
        src = DataSourceRelation(schema, batches)
        out = ctypes.c_void_p()
        err = _errbuf()
        _check(_ffi.lib().dfx_table_from_stream(ctypes.byref(src._take_stream()), ctypes.byref(out), err, 1024), err)
        return DeviceTable(out.value)

    @staticmethod
    def synth(cols: Sequence[tuple], seed: int, row_begin: int, n_rows: int) -> "DeviceTable":
        """cols: (name, kind, column_id, p0, p1); same generator as the CPU oracle's orc_synth_fill."""
        names = [c[0].encode() for c in cols]
        arr = (_ffi.SynthColumnC * len(cols))()
        for i, (_, kind, cid, p0, p1) in enumerate(cols):
            arr[i].name = names[i]
            arr[i].kind, arr[i].column_id, arr[i].p0, arr[i].p1 = kind, cid, p0, p1
        out = ctypes.c_void_p()
        err = _errbuf()
        _check(_ffi.lib().dfx_table_synth(arr, len(cols), ctypes.c_uint64(seed), row_begin, n_rows,
                                          ctypes.byref(out), err, 1024), err)
        return DeviceTable(out.value)

    def num_rows(self) -> int:
        return _ffi.lib().dfx_table_num_rows(self._h)

    def num_columns(self) -> int:
        return _ffi.lib().dfx_table_num_columns(self._h)

    def column_device_ptr(self, i: int) -> int:
        return _ffi.lib().dfx_table_column_device_ptr(self._h, i) or 0

    def scan(self, batch_rows: int = 0, row_begin: int = 0, n_rows: int = -1) -> Relation:
        """DataSourceRelation over the table -- or over its rows [row_begin, row_begin + n_rows) (row_begin: a multiple of 64)"""
        return _TableScan(self, batch_rows, row_begin, n_rows)

    def __del__(self):
        try:
            if self._h:
                _ffi.lib().dfx_table_free(self._h)
                self._h = None
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------------
# library-level helpers
# ---------------------------------------------------------------------------------------------------
def init(device: int = 0) -> None:
    err = _errbuf()
    _check(_ffi.lib().dfx_init(device, err, 1024), err)


def device_info() -> dict:
    name = ctypes.create_string_buffer(256)
    ncu, hbm, wf = ctypes.c_int32(), ctypes.c_int64(), ctypes.c_int32()
    err = _errbuf()
    _check(_ffi.lib().dfx_device_info(name, 256, ctypes.byref(ncu), ctypes.byref(hbm), ctypes.byref(wf), err, 1024), err)
    return {"name": name.value.decode(), "compute_units": ncu.value, "hbm_bytes": hbm.value, "wavefront": wf.value}


def synchronize() -> None:
    err = _errbuf()
    _check(_ffi.lib().dfx_synchronize(err, 1024), err)


def set_option(key: str, value: int) -> None:
    if _ffi.lib().dfx_set_option(key.encode(), int(value)) != 0:
        raise ExecutionError(3, f"unknown option {key}")


def explain(rel: "Relation") -> str:
    """Physical plan of an operator tree of this library (dfx_relation_explain): one line per operator."""
    st = rel._live_stream()
    L = _ffi.lib()
    n = L.dfx_relation_explain(ctypes.addressof(st), None, 0)
    if n < 0:
        raise ExecutionError(3, "not a stream of this library")
    buf = ctypes.create_string_buffer(int(n) + 1)
    L.dfx_relation_explain(ctypes.addressof(st), buf, int(n) + 1)
    return buf.value.decode()


def drain_on_device(rel: "Relation"):
    """Measurement hook (dfx_relation_drain_device): run `rel` to the end keeping its batches on the device. -> (rows, batches)"""
    L = _ffi.lib()
    rows, batches = ctypes.c_int64(), ctypes.c_int64()
    err = _errbuf()
    _check(L.dfx_relation_drain_device(ctypes.byref(rel._live_stream()), ctypes.byref(rows), ctypes.byref(batches), err, 1024), err)
    return rows.value, batches.value


def counter_get(name: str) -> int:
    return int(_ffi.lib().dfx_counter_get(name.encode()))


def counter_reset() -> None:
    _ffi.lib().dfx_counter_reset()


def profile_enable(on: bool) -> None:
    _ffi.lib().dfx_profile_enable(1 if on else 0)


def profile_reset() -> None:
    _ffi.lib().dfx_profile_reset()


def profile_snapshot() -> List[dict]:
    L = _ffi.lib()
    out = []
    for i in range(L.dfx_profile_count()):
        name = ctypes.create_string_buffer(64)
        n, ms, b = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
        if L.dfx_profile_get(i, name, 64, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(b)) == 0 and n.value:
            out.append({"kernel": name.value.decode(), "launches": n.value, "total_ms": ms.value,
                        "algo_bytes": b.value})
    return out
