"""The bench line contract (driver + tier framing): bench.py's defaults and the line the last GPU evidence run printed
(profiles/r06/bench_cls.json, bench_seg.json; the round-5 lines where a round-6 one is not committed) -- keys, units, and the arithmetic
between its fields.  CPU only: nothing is launched."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINES = [os.path.join(ROOT, "profiles", r) for r in ("r06", "r05")]

DRIVER_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
               "dtype", "data", "config")


def _line(name, rounds=LINES):
    for d in rounds:
        path = os.path.join(d, name)
        if os.path.exists(path):
            return json.loads(open(path).read().strip().splitlines()[-1])
    pytest.skip(f"{name} not committed")


def test_defaults_are_one_gpu_and_a_run_of_minutes(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert (a.gpus, a.workload, a.dtype, a.batch, a.points) == (1, "cls", "fp32", 32, 1024)      # configs[1], the metric's configuration
    assert 1 <= a.steps <= 200 and 0 <= a.warmup <= 50
    assert not a.no_graph and not a.no_pipeline and not a.no_optim          # nothing skipped inside the timed region by default


@pytest.mark.parametrize("name,batch", [("bench_cls.json", 32), ("bench_seg.json", 16)])
def test_committed_line_keeps_the_contract(name, batch):
    d = _line(name)
    for k in DRIVER_KEYS:
        assert k in d, k
    assert d["unit"] == "clouds/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["n_gpus"] == 1
    assert "workload" in d["config"] and "model" not in d["config"]
    published = json.load(open(os.path.join(ROOT, "BASELINE.json"))).get("published") or {}
    if not published:
        assert d["vs_baseline"] is None                                      # no published number for this metric: never a made-up ratio
    # value is whole-job throughput of exactly the timed steps
    assert d["steps_timed"] >= d["steps"]
    assert abs(d["value"] - batch * d["n_gpus"] / (d["ms_per_step"] * 1e-3)) <= 2e-3 * d["value"]
    assert d["dtype"] == "f32" and "arithmetic" in d                         # fp32 results; how the products are formed is said next to it
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 2e-3
    assert 0.0 < r["frac"] < 1.0
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["unit"] == d["unit"] and c["cores"] >= 1
    assert abs(d["gpu_over_cpu"] - d["value"] / c["value"]) <= 0.01 * d["gpu_over_cpu"]


def test_classification_line_names_the_metric_and_the_ball_query_roofline():
    d = _line("bench_cls.json")
    metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    assert metric.split(",")[0] in d["metric"]                               # "point-clouds/sec fwd+bwd"
    assert "configs[1]" in d["config"]["workload"] and d["config"]["optimizer_step"] is True
    assert d["fp32_mfma_ms_per_step"] > 0                                    # the same step on the fp32 MFMA instances, beside the headline
    b = d["roofline_ballquery"]
    assert b["bound"] == "hbm" and b["peak"] == 8000.0
    for clouds, rec in b["clouds_per_launch"].items():
        assert abs(rec["frac"] - rec["achieved"] / b["peak"]) <= 2e-3
        assert abs(rec["achieved"] - rec["algorithmic_bytes"] / (rec["us"] * 1e-6) / 1e9) <= 1e-2 * rec["achieved"]


def test_round6_line_carries_the_in_step_roofline_and_the_bracketing_legs():
    """VERDICT r5 items 1d / 3 / 4 on the driver's line: `roofline.frac` is the dominant class INSIDE the step's family graph with the
    optimistic alone-replay beside it; the dense / real-scan / eager throughputs and the dense GPU : dense CPU ratio; the ball query's
    build / query split."""
    d = _line("bench_cls.json", LINES[:1])
    r = d["roofline"]
    assert r["kernel"].startswith("rs_mlp_") and r["unit"] == "TFLOP/s"
    assert "alone_frac" in r and "alone_avg_launch_us" in r and r["stamp_gap_us"] > 0
    assert r["alone_frac"] >= r["frac"] > 0                                  # the warm-cache alone-replay is the optimistic one
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 2e-3
    amount = 2.0 * r["dims"][0] * r["dims"][1] * r["dims"][2]
    assert abs(r["achieved"] - amount / (r["avg_launch_us"] * 1e-6) / 1e12) <= 2e-2 * r["achieved"]
    for k in ("dense_clouds_per_s", "real_scans_clouds_per_s", "eager_clouds_per_s", "gpu_over_cpu_dense"):
        assert d.get(k) is not None and d[k] > 0, k
    assert d["dense_clouds_per_s"] < d["value"] and d["eager_clouds_per_s"] < d["dense_clouds_per_s"]
    assert abs(d["gpu_over_cpu_dense"] - d["dense_clouds_per_s"] / d["cpu_baseline"]["value"]) <= 0.01 * d["gpu_over_cpu_dense"]
    g = d["roofline_ballquery"]["clouds_per_launch"]["2048"]["prebuilt_grid"]
    assert abs(g["query_frac"] - g["query_algorithmic_bytes"] / (g["query_us"] * 1e-6) / 1e9 / 8000.0) <= 2e-3
    assert g["query_us"] < d["roofline_ballquery"]["clouds_per_launch"]["2048"]["us"] < g["build_plus_query_us"] * 1.2


def test_bench_parses_the_round6_flags(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py", "--workload", "seg", "--ragged", "--no-extra-legs"])
    a = bench.parse()
    assert a.ragged and a.no_extra_legs and a.workload == "seg"
