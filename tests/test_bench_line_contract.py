"""The bench line the driver parses (ROUND contract: one JSON line from rank 0): the committed FULL object of the final tree,
profiles/r06_bench_full.json (bench.py --full-out; the line itself is its compact form, profiles/r06_bench_line.json), carries every field of the contract with consistent values -- metric / unit of BASELINE.json,
whole-job value = rows / time, the roofline object of the dominant kernel (frac = achieved / peak, algorithmic bytes, PMC
traffic), the CPU baseline of the oracle port -- and every extra leg that claims a roofline also says whether it was checked
against the oracle.  CPU only: it guards the shape of what bench.py prints, not the numbers."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    with open(os.path.join(ROOT, "profiles", name)) as fh:
        return json.loads(fh.read().strip().splitlines()[-1])


def test_committed_bench_line_has_the_contract_fields():
    d = _line("r06_bench_full.json")
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert base["metric"].startswith(d["metric"]) and d["unit"] == "rows/s"  # BASELINE.json's metric (its qualifiers are the roofline object and --gpus)
    for key in ("value", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    rows = float(d["config"].get("rows_per_gpu", d["config"].get("rows", 0)) or 1e9)
    assert abs(d["value"] - rows / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]  # whole-job throughput = rows / step time
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] > 0
    for key in ("end_to_end_frac", "cold_first_step_ms", "avg_launch_ms", "frac_is"):
        assert key in r, key
    assert 0.0 < r["end_to_end_frac"] <= r["frac"] < 1.0  # the whole step cannot beat its dominant kernel
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert d["extra"]["verified_vs_oracle"]["ok"] is True


def test_every_extra_leg_with_a_roofline_says_whether_it_was_checked():
    d = _line("r06_bench_full.json")
    unchecked = []
    for name, leg in d["extra"].items():
        if not isinstance(leg, dict) or "roofline" not in leg:
            continue
        assert 0.0 < leg["roofline"]["frac"] < 1.0, name
        v = leg.get("verified_vs_oracle")
        if v is None:
            unchecked.append(name)
        else:
            assert (v.get("ok") if isinstance(v, dict) else v) is True, name
    # the two legs that re-time an already checked query with other options carry no second check
    assert set(unchecked) <= {"cfg2_filter_two_pass", "host_streamed_staged_ring"}, unchecked
    assert d["extra"]["csv_ingest_1gb"]["csv_general_tiles"] == 0  # the whole file went through the wave-cooperative path


def test_eight_rank_dry_run_line_used_the_library_exchange():
    d = _line("r06_bench_full_8rank_dryrun_one_gpu.json")
    assert d["n_gpus"] == 8 and d["config"]["rccl_ranks"] == 8
    assert "phases_ms" in d["extra"] and d["extra"]["verified_sum_of_group_sums_equals_ungrouped_sum"] is True
    ph = d["extra"]["phases_ms"]
    assert ph["collective_rounds"] == 2.0 and ph["host_syncs"] == 2.0  # round 6: the all-gather of states + counts, the buckets
    small = _line("r06_bench_line_8rank_dryrun_one_gpu.json")  # what that run printed last
    assert small["n_gpus"] == 8 and small["extra"]["phases_ms"]["collective_rounds"] == 2.0


def _check_contract(d):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline"):
        assert key in d, key
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "end_to_end_frac", "avg_launch_ms"):
        assert key in r, key
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert "workload" in d["config"] and "model" not in d["config"]


def test_the_line_bench_py_prints_last_is_compact():
    """Round 5's line was 20 KB and the driver could not parse it (BENCH_r05.json: parsed null).  The LAST stdout line is now
    built by bench.compact_line from the full object: the contract's fields, roofline, cpu_baseline and one {frac, ms, ok}
    per leg, under 4 KB whatever the full object holds.  Checked on bench.py's own line builder, not on a committed file."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    full = _line("r05_bench_line_final_commit.json")
    assert len(json.dumps(full)) > 16000  # the object that failed to parse
    small = bench.compact_line(full)
    text = json.dumps(small, separators=(",", ":"))
    assert len(text) < 4096 and "\n" not in text
    assert json.loads(text) == small  # strict JSON round trip (no NaN / Infinity)
    _check_contract(small)
    for key in ("metric", "value", "unit", "ms_per_step", "n_gpus", "steps", "warmup", "scaling", "dtype"):
        assert small[key] == full[key], key
    assert small["roofline"]["frac"] == full["roofline"]["frac"] and small["cpu_baseline"]["value"] == full["cpu_baseline"]["value"]
    legs = small["extra"]["legs"]
    for name in ("rows_1e10", "cfg3_groupby_sum_no_filter", "cfg5_q1_shape", "headline_selectivity_50", "headline_selectivity_80"):
        assert set(legs[name]) >= {"frac", "ok"} and legs[name]["ok"] is True, name
    # a pathological full object (hundreds of legs, long strings) still yields a line under the limit with every contract field
    fat = json.loads(json.dumps(full))
    for i in range(400):
        fat["extra"][f"leg_{i}"] = {"roofline": {"frac": 0.1}, "ms": 1.0, "verified_vs_oracle": {"ok": True}, "what": "x" * 500}
    fat["roofline"]["kernel"] = "k" * 5000
    fat["cpu_baseline"]["sample"] = "s" * 5000
    small2 = bench.compact_line(fat)
    assert len(json.dumps(small2, separators=(",", ":"))) < 4096
    _check_contract(small2)
    assert small2["extra"]["legs_dropped"] > 0


def test_emit_line_prints_the_compact_line_last(tmp_path, capsys, monkeypatch):
    import sys
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    full = _line("r05_bench_line_final_commit.json")
    bench.emit_line(full)
    out = capsys.readouterr()
    last = out.out.strip().splitlines()[-1]
    assert len(out.out.strip().splitlines()) == 1 and len(last) < 4096
    _check_contract(json.loads(last))
    assert json.loads(open(tmp_path / "gpurun_out" / "bench_extra.json").read()) == full  # the full object is kept beside it


def test_the_committed_compact_line_is_what_the_driver_gets():
    d = _line("r06_bench_line.json")
    assert len(json.dumps(d, separators=(",", ":"))) < 4096
    _check_contract(d)
    full = _line("r06_bench_full.json")
    assert d["value"] == full["value"] and d["roofline"]["frac"] == full["roofline"]["frac"]
    assert d["extra"]["scaling_anchor_rows_per_s"] == round(full["extra"]["rows_1e10"]["rows_per_s"], 3)
    assert all(leg.get("ok", True) is True for leg in d["extra"]["legs"].values())
    assert full["extra"]["csv_ingest_1gb"]["verified_vs_oracle"]["every_record"]["columns_bit_exact"] is True
