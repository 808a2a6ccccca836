"""CPU: the committed bench lines (profiles/r0N_bench_*.json, written by bench.py
on the MI355X) carry every field of the driver's contract, and their derived
numbers are consistent with each other."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINES = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[2-9]_bench_*.json")))
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


def test_round3_driver_line():
    """Round 3: fresh actions every step, traffic and issue counters of the same
    round, the sort node broken down by chain, a step traffic figure that counts
    the step's own kernels only."""
    d = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_default.json")))
    assert d["n_gpus"] == 1 and d["config"]["worlds_per_gpu"] == 8192
    assert "configs[2]" in d["config"]["workload"]
    assert "every step" in d["data"]
    nodes = d["roofline"]["nodes"]
    assert {"sort_node", "physics_step", "physics_step_issue", "step"} <= set(nodes)
    for key in ("sort_node", "physics_step", "step"):
        assert nodes[key]["traffic_source"] == "profiles/r03_hbm_traffic.json", key
    chains = nodes["sort_node"]["chains"]
    assert len(chains) >= 2
    assert sum(c["avg_us"] for c in chains) == pytest.approx(
        nodes["sort_node"]["avg_us"], rel=1e-3)
    assert sum(c["algo_bytes"] for c in chains) == pytest.approx(
        nodes["sort_node"]["algo_bytes_per_launch"], rel=1e-6)
    # PMC bytes per step: above the algorithmic figure of its dominant kernels,
    # nowhere near what the bench's own bandwidth probe moves
    assert nodes["physics_step"]["traffic"] < nodes["step"]["traffic"] < 2e9
    issue = nodes["physics_step_issue"]
    assert issue["bound"] == "valu-issue" and 0 < issue["frac"] < 1
    assert d["roofline"]["peak_measured"]["copy_GBps"] > 1000


def test_step_traffic_counts_the_step_kernels_only():
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "make_traffic_json", os.path.join(ROOT, "profiles", "tools", "make_traffic_json.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    entries = [
        {"kernel": "physics:worldStep(LDS)", "launches_per_step": 1.0, "traffic_bytes": 200},
        {"kernel": "SortArchetype:sort.gather", "launches_per_step": 1.97, "traffic_bytes": 100},
        {"kernel": "SortArchetype:sort.histogram", "launches_per_step": 0.0, "traffic_bytes": 7},
        {"kernel": "at::native::vectorized_elementwise_kernel<4, ...>",
         "launches_per_step": 2.2, "traffic_bytes": 10 ** 9},
        {"kernel": "__amd_rocclr_copyBuffer", "launches_per_step": 3.4, "traffic_bytes": 10 ** 9},
        {"kernel": "madrona::mwGPU::entryKernels::initWorlds<...>",
         "launches_per_step": 0.6, "traffic_bytes": 10 ** 6},
    ]
    step = m.step_entry("sim", 8, entries)
    assert step["kernel"] == "step:all-kernels" and step["traffic_bytes"] == 300
    assert m.bench_name("void madrona::mwhip::(anonymous namespace)::sortCompactPrepare(...)") \
        == "SortArchetype:sort.compact.prepare"
    assert m.bench_name("madrona::mwhip::renderRaycast<true>") == "render:raycast"
