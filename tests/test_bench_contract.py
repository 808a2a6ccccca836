"""CPU: the committed bench lines (profiles/r02_bench_*.json, written by bench.py
on the MI355X) carry every field of the driver's contract, and their derived
numbers are consistent with each other."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINES = sorted(glob.glob(os.path.join(ROOT, "profiles", "r02_bench_*.json")))
LINES = [p for p in LINES if "under_rocprof" not in p]

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
            "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline"]


@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_committed_bench_line_has_the_contract_fields(path):
    d = json.load(open(path))
    for key in REQUIRED:
        assert key in d, key
    assert d["unit"] == "steps/s" and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert "workload" in d["config"] and "BASELINE.json configs" in d["config"]["workload"]
    assert "model" not in d["config"]
    # value = worlds x steps / time, ms_per_step = time / steps
    worlds = d["config"]["total_worlds"]
    assert d["value"] == pytest.approx(worlds / (d["ms_per_step"] * 1e-3), rel=1e-6)
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], abs=2e-4)
    assert 0 < r["frac"] < 1
    if d.get("cpu_baseline") is not None:
        c = d["cpu_baseline"]
        for key in ("value", "unit", "cores", "kind", "sample"):
            assert key in c, key
        assert c["kind"] in ("reference", "port") and c["cores"] >= 1


def test_the_driver_line_is_the_configuration_the_metric_is_quoted_on():
    d = json.load(open(os.path.join(ROOT, "profiles", "r02_bench_default.json")))
    assert d["n_gpus"] == 1 and d["config"]["worlds_per_gpu"] == 8192
    assert "configs[2]" in d["config"]["workload"]
    assert d["cpu_baseline"]["kind"] == "reference"
    nodes = d["roofline"]["nodes"]
    assert {"sort_node", "physics_step", "step"} <= set(nodes)
    # the sort node is every kernel of the chain, not its best one
    assert nodes["sort_node"]["launches"] >= 4
    assert d["ecs_config2"]["value"] > 0 and d["render_config5"]["value"] > 0
