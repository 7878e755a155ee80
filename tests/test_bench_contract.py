"""The bench.py JSON line contract (driver-facing) checked on the committed round profile: every key the driver and
the judge read is present with the right type, and the numbers are mutually consistent."""
import json
import os

import pytest

from conftest import ROOT


def _line(name):
    path = os.path.join(ROOT, "profiles", "r01", name)
    if not os.path.exists(path):
        pytest.skip(name + " not committed")
    return json.loads(open(path).read().strip().splitlines()[-1])


def test_headline_line_has_the_contract_keys():
    j = _line("bench_c4_n1.json")
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(j[k], t), k
    assert "vs_baseline" in j and j["vs_baseline"] is None      # BASELINE.md publishes no number for this metric
    assert j["n_gpus"] == 1 and j["higher_is_better"] is True and j["dtype"] == "f32" and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"]
    W, H, D = j["config"]["W"], j["config"]["H"], j["config"]["D"]
    assert (W, H, D) == (1920, 1080, 256)
    assert abs(j["value"] - 2.0 * W * H * D / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-6
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["alg_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-3
    assert r["traffic"] is None or r["traffic"] > 0
    c = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["unit"] == j["unit"]


def test_distributed_lines_were_checked_against_the_single_gpu_run():
    for name in ("bench_c4_dist_world1.json", "bench_c4_dist_world1_allgather.json", "bench_c4_dist_world1_nooverlap.json"):
        j = _line(name)
        assert j["verified_vs_single_gpu"] is True, name
        assert j["scaling"] == "strong"
