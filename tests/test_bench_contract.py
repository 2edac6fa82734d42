"""The line bench.py prints, against the driver's contract (no GPU here: the committed line of the last profiled default run,
profiles/r04_bench_default.json, stands in for a fresh one; tests/test_gpu_full_size.py runs bench.py itself)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    with open(os.path.join(ROOT, "profiles", "r04_bench_default.json")) as fh:
        return json.loads([ln for ln in fh if ln.startswith("{")][-1])


def test_headline_fields():
    d = _line()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert "64x512x512" in d["config"]["workload"] and "1024" in d["config"]["workload"]
    # value = evaluations of all steps / wall time of the timed region
    evals_per_step = 512 * 512 * 1024 * 64
    assert abs(d["value"] - evals_per_step / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert d["roofline"]["kernel_ms"] <= d["ms_per_step"]


def test_roofline_block_names_what_binds_and_its_clock():
    r = _line()["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "compulsory_bytes", "traffic_over_compulsory",
                "frac_algorithmic", "kernel", "kernel_ms", "note"):
        assert key in r, key
    assert r["bound"] in ("lds+issue", "hbm", "mfma") and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] < 1.0
    assert r["bound"] == "lds+issue" and r["cache_resident"] is True and "LDS bytes" in r["note"] and "kernel_ms" in r["note"]
    # the algorithmic bytes of SURVEY 8(d) are the bytes the LDS rate is made of; against HBM they exceed the peak and are labelled
    assert r["frac_algorithmic"] > 1.0 and abs(r["lds_read_GBps"] - r["achieved"]) < 1e-6
    assert r["traffic"] is not None and r["traffic_source"].startswith("live")          # measured by the run itself
    assert abs(r["traffic_over_compulsory"] - r["traffic"] / r["compulsory_bytes"]) < 1e-9
    assert r["env_overrides"] == 0


def test_baselines_and_masked_entry():
    d = _line()
    port, full = d["cpu_baseline"], d["cpu_baseline_full_sort"]
    assert port["kind"] == "port" and full["kind"] == "restatement" and port["cores"] >= 1 and full["cores"] >= 1
    assert port["unit"] == full["unit"] == "evals/s" and "sample" in port and "search_cpu_only" in full["sample"]
    m = d["masked"]
    assert m["mask_fraction"] == 0.01 and m["kernel_ms"] <= m["ms_per_step"] and m["ms_per_step"] > d["ms_per_step"] * 0.9
