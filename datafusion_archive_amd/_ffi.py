"""ctypes declarations of the C ABI (include/dfx.h) and the loader of the HIP library.

There is no CPU fallback: if ``libdfx_hip.so`` cannot be built or loaded the import fails loudly.
"""
from __future__ import annotations

import ctypes
import os
import sys

from .logicalplan import ExprNode

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libdfx_hip.so")


class ArrowSchema(ctypes.Structure):
    pass


ArrowSchema._fields_ = [
    ("format", ctypes.c_char_p), ("name", ctypes.c_char_p), ("metadata", ctypes.c_char_p),
    ("flags", ctypes.c_int64), ("n_children", ctypes.c_int64),
    ("children", ctypes.POINTER(ctypes.POINTER(ArrowSchema))), ("dictionary", ctypes.POINTER(ArrowSchema)),
    ("release", ctypes.c_void_p), ("private_data", ctypes.c_void_p),
]


class ArrowArray(ctypes.Structure):
    pass


ArrowArray._fields_ = [
    ("length", ctypes.c_int64), ("null_count", ctypes.c_int64), ("offset", ctypes.c_int64),
    ("n_buffers", ctypes.c_int64), ("n_children", ctypes.c_int64),
    ("buffers", ctypes.POINTER(ctypes.c_void_p)), ("children", ctypes.POINTER(ctypes.POINTER(ArrowArray))),
    ("dictionary", ctypes.POINTER(ArrowArray)), ("release", ctypes.c_void_p), ("private_data", ctypes.c_void_p),
]


class ArrowArrayStream(ctypes.Structure):
    pass


GET_SCHEMA = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.POINTER(ArrowArrayStream), ctypes.POINTER(ArrowSchema))
GET_NEXT = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.POINTER(ArrowArrayStream), ctypes.POINTER(ArrowArray))
GET_LAST_ERROR = ctypes.CFUNCTYPE(ctypes.c_char_p, ctypes.POINTER(ArrowArrayStream))
RELEASE_STREAM = ctypes.CFUNCTYPE(None, ctypes.POINTER(ArrowArrayStream))

ArrowArrayStream._fields_ = [
    ("get_schema", GET_SCHEMA), ("get_next", GET_NEXT), ("get_last_error", GET_LAST_ERROR),
    ("release", RELEASE_STREAM), ("private_data", ctypes.c_void_p),
]


class OptionC(ctypes.Structure):
    _fields_ = [("key", ctypes.c_char_p), ("value", ctypes.c_int64)]


class SynthColumnC(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("kind", ctypes.c_int32), ("column_id", ctypes.c_int32),
                ("p0", ctypes.c_double), ("p1", ctypes.c_double)]


# every symbol include/dfx.h declares (the CPU test-suite checks the library exports all of them)
EXPORTED_SYMBOLS = [
    "dfx_abi_version", "dfx_init", "dfx_device_info", "dfx_synchronize",
    "dfx_compile_scalar_expr", "dfx_compile_expr", "dfx_runtime_expr_name", "dfx_runtime_expr_type",
    "dfx_runtime_expr_is_aggregate", "dfx_runtime_expr_free",
    "dfx_filter_relation_new", "dfx_project_relation_new", "dfx_aggregate_relation_new",
    "dfx_filter_relation_new_with_options", "dfx_aggregate_relation_new_with_options",
    "dfx_table_from_stream", "dfx_table_synth", "dfx_table_num_rows", "dfx_table_num_columns",
    "dfx_table_column_device_ptr", "dfx_table_scan_new", "dfx_table_scan_range_new", "dfx_table_free", "dfx_csv_datasource_new",
    "dfx_sort_relation_new", "dfx_limit_relation_new",
    "dfx_aggregate_partial_build", "dfx_aggregate_partial_export", "dfx_aggregate_partial_import",
    "dfx_comm_unique_id", "dfx_comm_init", "dfx_comm_destroy", "dfx_comm_ranks", "dfx_aggregate_exchange",
    "dfx_profile_enable", "dfx_profile_reset", "dfx_profile_count", "dfx_profile_get", "dfx_set_option",
    "dfx_counter_get", "dfx_counter_reset", "dfx_relation_explain", "dfx_relation_drain_device", "dfx_filter_debug_mask",
    "dfx_debug_group_hash", "dfx_debug_unhash32", "dfx_debug_plan_term",
]

_lib = None


def build_library(force: bool = False) -> str:
    from . import build as _build

    return _build.build(force=force)


def lib() -> ctypes.CDLL:
    """Load (building it first if it is stale or missing) the HIP library. Raises on failure."""
    global _lib
    if _lib is not None:
        return _lib
    if os.environ.get("DFX_NO_TORCH") != "1":
        # torch bundles its own libamdhip64.so.7 (same soname as /opt/rocm's).  Importing it first
        # makes this process use ONE HIP runtime for torch.distributed/RCCL plumbing and for dfx.
        try:
            import torch  # noqa: F401
        except Exception:  # torch is optional plumbing; the library itself does not need it
            pass
    try:
        # DFX_LIB: load this (already built) variant of the library instead -- A/B runs of kernel build options
        path = os.environ.get("DFX_LIB") or build_library()
    except Exception as e:  # stale/missing library and no compiler: fail loudly, no fallback
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"datafusion_archive_amd: the HIP extension {LIB_PATH} is missing and could "
                              f"not be built ({e}). There is no CPU fallback.") from e
        path = LIB_PATH
    L = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    c_err = [ctypes.c_char_p, ctypes.c_size_t]
    P = ctypes.POINTER
    L.dfx_abi_version.restype = ctypes.c_int32
    L.dfx_init.argtypes = [ctypes.c_int32] + c_err
    L.dfx_device_info.argtypes = [ctypes.c_char_p, ctypes.c_size_t, P(ctypes.c_int32), P(ctypes.c_int64),
                                  P(ctypes.c_int32)] + c_err
    L.dfx_synchronize.argtypes = c_err
    for fn in (L.dfx_compile_scalar_expr, L.dfx_compile_expr):
        fn.argtypes = [P(ExprNode), ctypes.c_int32, ctypes.c_int32, P(ArrowSchema), P(ctypes.c_void_p)] + c_err
        fn.restype = ctypes.c_int32
    L.dfx_runtime_expr_name.argtypes = [ctypes.c_void_p]
    L.dfx_runtime_expr_name.restype = ctypes.c_char_p
    L.dfx_runtime_expr_type.argtypes = [ctypes.c_void_p]
    L.dfx_runtime_expr_is_aggregate.argtypes = [ctypes.c_void_p]
    L.dfx_runtime_expr_free.argtypes = [ctypes.c_void_p]
    L.dfx_runtime_expr_free.restype = None
    L.dfx_filter_relation_new.argtypes = [P(ArrowArrayStream), ctypes.c_void_p, P(ArrowSchema),
                                          P(ArrowArrayStream)] + c_err
    L.dfx_filter_relation_new_with_options.argtypes = [P(ArrowArrayStream), ctypes.c_void_p, P(ArrowSchema), P(OptionC), ctypes.c_int32,
                                                       P(ArrowArrayStream)] + c_err
    L.dfx_aggregate_relation_new_with_options.argtypes = [P(ArrowSchema), P(ArrowArrayStream), P(ctypes.c_void_p), ctypes.c_int32,
                                                          P(ctypes.c_void_p), ctypes.c_int32, P(OptionC), ctypes.c_int32,
                                                          P(ArrowArrayStream)] + c_err
    L.dfx_project_relation_new.argtypes = [P(ArrowArrayStream), P(ctypes.c_void_p), ctypes.c_int32,
                                           P(ArrowSchema), P(ArrowArrayStream)] + c_err
    L.dfx_aggregate_relation_new.argtypes = [P(ArrowSchema), P(ArrowArrayStream), P(ctypes.c_void_p),
                                             ctypes.c_int32, P(ctypes.c_void_p), ctypes.c_int32,
                                             P(ArrowArrayStream)] + c_err
    L.dfx_table_from_stream.argtypes = [P(ArrowArrayStream), P(ctypes.c_void_p)] + c_err
    L.dfx_table_synth.argtypes = [P(SynthColumnC), ctypes.c_int32, ctypes.c_uint64, ctypes.c_int64,
                                  ctypes.c_int64, P(ctypes.c_void_p)] + c_err
    L.dfx_table_num_rows.argtypes = [ctypes.c_void_p]
    L.dfx_table_num_rows.restype = ctypes.c_int64
    L.dfx_table_num_columns.argtypes = [ctypes.c_void_p]
    L.dfx_table_column_device_ptr.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    L.dfx_table_column_device_ptr.restype = ctypes.c_void_p
    L.dfx_table_scan_new.argtypes = [ctypes.c_void_p, ctypes.c_int64, P(ArrowArrayStream)] + c_err
    L.dfx_table_scan_range_new.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, P(ArrowArrayStream)] + c_err
    L.dfx_table_free.argtypes = [ctypes.c_void_p]
    L.dfx_table_free.restype = None
    L.dfx_sort_relation_new.argtypes = [P(ArrowArrayStream), P(ctypes.c_void_p), P(ctypes.c_int32), ctypes.c_int32,
                                        P(ArrowSchema), P(ArrowArrayStream)] + c_err
    L.dfx_limit_relation_new.argtypes = [P(ArrowArrayStream), ctypes.c_int64, P(ArrowSchema), P(ArrowArrayStream)] + c_err
    L.dfx_csv_datasource_new.argtypes = [ctypes.c_char_p, P(ArrowSchema), ctypes.c_int64, P(ArrowArrayStream)] + c_err
    L.dfx_aggregate_partial_build.argtypes = [P(ArrowArrayStream), ctypes.c_int32, P(ctypes.c_int32),
                                              P(ctypes.c_int64)] + c_err
    L.dfx_aggregate_partial_export.argtypes = [P(ArrowArrayStream), ctypes.c_void_p, ctypes.c_int64] + c_err
    L.dfx_aggregate_partial_import.argtypes = [P(ArrowArrayStream), ctypes.c_void_p, P(ctypes.c_int64),
                                               ctypes.c_int32] + c_err
    L.dfx_comm_unique_id.argtypes = [ctypes.c_char_p] + c_err
    L.dfx_comm_init.argtypes = [ctypes.c_char_p, ctypes.c_int32, ctypes.c_int32, P(ctypes.c_void_p)] + c_err
    L.dfx_comm_destroy.argtypes = [ctypes.c_void_p]
    L.dfx_comm_destroy.restype = None
    L.dfx_comm_ranks.argtypes = [ctypes.c_void_p]
    L.dfx_comm_ranks.restype = ctypes.c_int32
    L.dfx_aggregate_exchange.argtypes = [P(ArrowArrayStream), ctypes.c_void_p, P(ctypes.c_int64)] + c_err
    L.dfx_profile_enable.argtypes = [ctypes.c_int32]
    L.dfx_profile_get.argtypes = [ctypes.c_int32, ctypes.c_char_p, ctypes.c_size_t, P(ctypes.c_int64),
                                  P(ctypes.c_double), P(ctypes.c_double)]
    L.dfx_set_option.argtypes = [ctypes.c_char_p, ctypes.c_int64]
    L.dfx_relation_explain.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    L.dfx_relation_explain.restype = ctypes.c_int64
    L.dfx_relation_drain_device.argtypes = [ctypes.c_void_p, P(ctypes.c_int64), P(ctypes.c_int64)] + c_err
    L.dfx_filter_debug_mask.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, P(ctypes.c_int64)] + c_err
    L.dfx_debug_group_hash.argtypes = [ctypes.c_uint64]
    L.dfx_debug_group_hash.restype = ctypes.c_uint64
    L.dfx_debug_unhash32.argtypes = [ctypes.c_uint32]
    L.dfx_debug_unhash32.restype = ctypes.c_uint32
    L.dfx_debug_plan_term.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int32]
    L.dfx_debug_plan_term.restype = ctypes.c_int32
    L.dfx_counter_get.argtypes = [ctypes.c_char_p]
    L.dfx_counter_get.restype = ctypes.c_int64
    L.dfx_counter_reset.restype = None
    _lib = L
    return L
