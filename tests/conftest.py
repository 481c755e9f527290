"""pytest configuration: registers the `gpu` marker; everything unmarked must pass on CPU."""
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# every option of the library that a GPU test may have changed goes back to its default afterwards (a failing assertion in
# the middle of a test must not leak, say, a forced GROUP BY strategy into the tests that follow)
_OPTION_DEFAULTS = {
    "agg.strategy": 0, "agg.capacity_log2": 0, "agg.lds_slots": -1, "agg.lds_copies": -1, "scan.fast": 1,
    "agg.partition_mode": 2, "agg.partition_block": 1024, "agg.fewgroup": 1, "agg.replay_in_place": 1,
    "agg.partition_pad": 0, "agg.dict_capacity_log2": 0, "agg.partition_cap_rows": 0, "agg.partition_defer": 0, "agg.partition_split_rows": 1 << 26,
    "agg.partition_defer_batches": 8, "agg.pass2_stream": 1, "agg.calibration_memo": 1, "agg.emit_async": 1,
    "agg.hot_keys": -1, "agg.partition_layout": 1, "agg.partition_producers": 0, "agg.narrow_keys": -1,
    "agg.ctrl_snapshot": 1, "filter.single_pass": 1, "agg.pass1_ws": 8, "agg.pass1_ws_dense": 0, "agg.pass1_ws_dense_scanners": 4,
    # tests scan resident tables in SMALL batches on purpose (multi-batch paths, deferred pass 2 ...): the library's default
    # (one slice per routing window for an aggregate over a table scan) would merge them away
    "agg.merge_scan_batches": 0, "agg.early_keys": 1, "csv.wave_tiles": 1, "export.kernel_copy": 1, "agg.narrow_chunk16": 1, "agg.shared_operand": 1, "agg.chunk_hold": 4, "agg.pair_scan": 1, "agg.shared_planes": 1,
}


@pytest.fixture(autouse=True)
def _restore_library_options(request):
    if request.node.get_closest_marker("gpu") is not None:
        from datafusion_archive_amd import execution as ex
        ex.set_option("agg.merge_scan_batches", 0)
    yield
    if request.node.get_closest_marker("gpu") is not None:
        from datafusion_archive_amd import execution as ex
        for k, v in _OPTION_DEFAULTS.items():
            ex.set_option(k, v)
