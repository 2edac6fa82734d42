"""The line bench.py prints, against the driver's contract.

Two kinds of test, kept apart on purpose:
 * `test_assembly_*` call the functions that ASSEMBLE the line (bench.roofline_block, bench.headline_line) with stubbed
   measurements -- no GPU, no library -- so a regression in bench.py's arithmetic or field set fails here;
 * `test_artefact_*` are artefact-consistency checks of the committed line of the last profiled default run under profiles/
   (what a reader of the repository sees); they say nothing about bench.py's code.  tests/test_gpu_full_size.py runs bench.py
   itself on the GPU box."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _stub_roofline(bench, **over):
    evals = 512 * 512 * 1024 * 64
    kw = dict(kernel_ms=2.0, algorithmic_bytes=evals * 8 + 58_000_000, lds_read_bytes=evals * 8, evals_per_step=evals,
              traffic=13_000_000_000, traffic_source="live: stub", traffic_rejected=None, compulsory_bytes=193_000_000,
              padded_copy_bytes_guess=134 << 20, is_lds_kernel=True, kernel="kb::kb_search_lds<8, 16, 16, 4, true, false, 3>",
              edge_count_tables=1, mask_fraction=0.0, padded_copy_reused=1, env_overrides=0, copy_gbps=6300.0, read_gbps=7100.0,
              lds_gbps=127000.0, psi_phi_bytes=134217728)
    kw.update(over)
    return bench.roofline_block(**kw), kw


def test_assembly_cache_resident_line_names_lds_and_one_clock():
    import bench

    r, kw = _stub_roofline(bench)
    assert r["bound"] == "lds+issue" and r["unit"] == "GB/s" and r["peak"] == bench.LDS_PEAK_GBPS and r["cache_resident"] is True
    # achieved = LDS bytes / kernel_ms; frac = achieved / peak; every other rate on the same clock
    want = kw["lds_read_bytes"] / 2.0e-3 / 1e9
    assert abs(r["achieved"] - want) < 1e-6 and abs(r["frac"] - want / bench.LDS_PEAK_GBPS) < 1e-12 and 0 < r["frac"] < 1
    assert abs(r["lds_read_GBps"] - r["achieved"]) < 1e-9
    assert abs(r["fabric_GBps"] - 13e9 / 2.0e-3 / 1e9) < 1e-6
    assert abs(r["kernel_evals_per_s"] - kw["evals_per_step"] / 2.0e-3) < 1.0
    # the algorithmic bytes against HBM exceed the peak and are labelled as what they are
    assert r["frac_algorithmic"] > 1.0 and "LDS bytes" in r["note"] and "kernel_ms" in r["note"]
    assert abs(r["traffic_over_compulsory"] - 13e9 / 193e6) < 1e-9
    assert abs(r["frac_of_measured_lds"] - want / 127000.0) < 1e-9
    assert "OUTSIDE the timed steps" in r["padded_copy"] and "tables" in r["obs_counts"]


def test_assembly_traced_values_make_the_fraction_recomputable():
    """roofline.traced_kernel_ms / lds_insts / lds_cycles (bench.live_traffic: the run's own rocprofv3 sub-runs): a reader
    recomputes `frac` on the tracer's clock and the one-ds_read_b64-per-64-evaluations identity from the line alone."""
    import bench

    evals = 512 * 512 * 1024 * 64
    traced = {"traced_kernel_ms": 2.1, "traced_launches": 12, "lds_insts": evals / 64 * 1.006, "lds_cycles": evals / 64 * 2.04}
    r, kw = _stub_roofline(bench, traced=traced)
    assert r["traced_kernel_ms"] == 2.1 and r["traced_launches"] == 12
    assert abs(r["frac_at_traced_kernel_ms"] - kw["lds_read_bytes"] / 2.1e-3 / 1e9 / bench.LDS_PEAK_GBPS) < 1e-12
    assert abs(r["evals_per_lds_inst"] - 64 / 1.006) < 1e-9 and abs(r["lds_cycles_per_inst"] - 2.04 / 1.006) < 1e-9
    # without the sub-runs (N > 1, --no-live-traffic) the keys are there and empty
    r0, _ = _stub_roofline(bench)
    for key in ("traced_kernel_ms", "traced_launches", "frac_at_traced_kernel_ms", "lds_insts", "lds_cycles", "evals_per_lds_inst",
                "lds_cycles_per_inst"):
        assert key in r0 and r0[key] is None
    # the LDS pass is optional: a byte-only measurement still carries the traced duration
    r1, _ = _stub_roofline(bench, traced={"traced_kernel_ms": 2.2, "traced_launches": 8})
    assert r1["traced_kernel_ms"] == 2.2 and r1["lds_insts"] is None and r1["evals_per_lds_inst"] is None
    json.dumps(r)


def test_assembly_hbm_resident_line_uses_fabric_bytes_or_says_lower_bound():
    import bench

    r, _ = _stub_roofline(bench, padded_copy_bytes_guess=17 << 30, traffic=88_000_000_000, kernel_ms=20.0, compulsory_bytes=21_000_000_000)
    assert r["bound"] == "hbm" and r["peak"] == bench.HBM_PEAK_GBPS and r["cache_resident"] is False
    assert abs(r["achieved"] - 88e9 / 20e-3 / 1e9) < 1e-6 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    r2, _ = _stub_roofline(bench, padded_copy_bytes_guess=17 << 30, traffic=None, traffic_source=None, kernel_ms=20.0,
                           compulsory_bytes=21_000_000_000)
    assert r2["traffic"] is None and r2["traffic_over_compulsory"] is None and r2["fabric_GBps"] is None
    assert "LOWER bound" in r2["note"] and abs(r2["achieved"] - 21e9 / 20e-3 / 1e9) < 1e-6
    # a direct-kernel run on a cache-resident array is not priced against LDS
    r3, _ = _stub_roofline(bench, is_lds_kernel=False, lds_read_bytes=0)
    assert r3["bound"] == "hbm" and r3["frac_of_guide_lds"] is None and r3["frac_of_measured_lds"] is None
    # masked stack / made-by-this-search wording
    r4, _ = _stub_roofline(bench, mask_fraction=0.01, padded_copy_reused=0)
    assert "counted per sample" in r4["obs_counts"] and "inside the step" in r4["padded_copy"]


def test_assembly_headline_fields_and_value():
    import bench

    roof, kw = _stub_roofline(bench)
    evals = kw["evals_per_step"]
    d = bench.headline_line(total_evals=evals * 20, elapsed_s=0.05, world=1, steps=20, warmup=3, dtype="f32",
                            config={"workload": "64x512x512 ... 1024 ..."}, roofline=roof)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "model" not in d["config"]
    assert abs(d["ms_per_step"] - 2.5) < 1e-12 and abs(d["value"] - evals / 2.5e-3) / d["value"] < 1e-12
    d8 = bench.headline_line(total_evals=evals * 8 * 20, elapsed_s=0.06, world=8, steps=20, warmup=3, dtype="f32", config={}, roofline=roof)
    assert d8["n_gpus"] == 8 and abs(d8["value"] - evals * 8 * 20 / 0.06) < 1.0   # whole-job aggregate, not per GPU
    json.dumps(d)   # the line is JSON


# ---- artefact consistency (the committed line of the last profiled default run; not a test of bench.py) ----
def _artefact():
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_bench_default.json")))
    if not found:
        pytest.skip("no committed default bench line under profiles/")
    with open(found[-1]) as fh:
        return json.loads([ln for ln in fh if ln.startswith("{")][-1])


def test_artefact_headline_line_is_consistent():
    d = _artefact()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert "64x512x512" in d["config"]["workload"] and "1024" in d["config"]["workload"]
    evals_per_step = 512 * 512 * 1024 * 64
    assert abs(d["value"] - evals_per_step / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["kernel_ms"] <= d["ms_per_step"] and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] < 1.0
    assert r["traffic"] is None or abs(r["traffic_over_compulsory"] - r["traffic"] / r["compulsory_bytes"]) < 1e-9


def test_artefact_baselines_and_masked_entry():
    d = _artefact()
    port, full = d["cpu_baseline"], d["cpu_baseline_full_sort"]
    assert port["kind"] == "port" and full["kind"] == "restatement" and port["cores"] >= 1 and full["cores"] >= 1
    assert port["unit"] == full["unit"] == "evals/s" and "sample" in port and "search_cpu_only" in full["sample"]
    m = d["masked"]
    assert m["mask_fraction"] == 0.01 and m["kernel_ms"] <= m["ms_per_step"]
