"""CPU: the bench's own arithmetic (rooflines from a kernel table, the PMC
traffic tool) on synthetic inputs, and a schema check of the ONE driver-shaped
line that is committed (the newest profiles/r0N_bench_default.json)."""
import glob
import importlib.util
import json
import os
import sqlite3
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _stat(name, us, algo, rows=1000.0, declared=False):
    return {"name": name, "kind": 0, "avg_us": us, "algo_bytes": algo, "rows": rows,
            "io_declared": declared, "workgroups": 1, "node_index": 0}


def test_parallel_for_and_sort_node_rooflines():
    """bench.rooflines: declared ParallelFor nodes get their own fraction (sum of
    declared bytes / sum of kernel times), nodes on the signature rule are kept
    apart, the sort node is priced both ways."""
    bench = _load("bench", os.path.join(ROOT, "bench.py"))
    stats = [
        _stat("sim::moveSystem", 5.0, 1.0e6, declared=True),
        _stat("sim::obsSystem", 15.0, 7.0e6, declared=True),
        _stat("sim::legacySystem", 10.0, 3.0e6),
        _stat("SortArchetype:sort.compact.prepare", 5.0, 16.0e6, rows=1.0e6),
        _stat("SortArchetype:sort.compact.scatter", 5.0, 16.0e6, rows=1.0e6),
        _stat("SortArchetype:sort.gather", 30.0, 208.0e6, rows=1.0e6),
    ]
    r = bench.rooflines(stats, "escape_room", 4096, 0.1)
    nodes = r["nodes"]
    pf = nodes["parallel_for"]
    assert pf["launches"] == 2 and pf["algo_bytes_per_launch"] == 8_000_000
    assert pf["achieved"] == pytest.approx(8.0e6 / 20e-6 / 1e9, rel=1e-3)
    assert pf["frac"] == pytest.approx(pf["achieved"] / 8000.0, abs=1e-4)
    assert [k["name"] for k in pf["kernels"]] == ["sim::moveSystem", "sim::obsSystem"]
    assert nodes["parallel_for_signature_rule"]["launches"] == 1
    sort = nodes["sort_node"]
    assert sort["launches"] == 3 and sort["algo_bytes_per_launch"] == 240_000_000
    # what the compaction chain moves: the gather + 16 B per row of keys / indices
    assert sort["bytes_moved_estimate"] == int(208.0e6 + 16.0 * 1.0e6)
    assert sort["frac_bytes_moved"] < sort["frac"]
    assert "compact.prepare" in sort["kernel"] and "histogram" not in sort["kernel"]
    both = nodes["sort_and_parallel_for"]
    assert both["launches"] == 5
    assert both["algo_bytes_per_launch"] == 248_000_000
    table = bench.kernel_table(stats, "escape_room", 4096)
    assert table[0]["bytes"] == "declared read/write set" and "GBps" in table[0]
    assert "algo_MB_signature_upper_bound" in table[2]


def _pmc_db(path, counter, per_kernel):
    db = sqlite3.connect(path)
    db.execute("create table counters_collection (kernel_name text, counter_name text, "
               "value real)")
    for name, values in per_kernel.items():
        for v in values:
            db.execute("insert into counters_collection values (?, ?, ?)",
                       (name, counter, v))
    db.commit()
    db.close()


def test_traffic_tool_normalises_each_pass_by_its_own_replays(tmp_path, capsys):
    """make_traffic_json: the FETCH_SIZE and WRITE_SIZE passes replay the step
    graph a different number of times (the bench repeats its window until it is
    long enough); round 3 divided both by the fetch pass's count and reported a
    1.5 x write amplification of the gather that was not there."""
    m = _load("make_traffic_json",
              os.path.join(ROOT, "profiles", "tools", "make_traffic_json.py"))
    fetch, write = str(tmp_path / "f.db"), str(tmp_path / "w.db")
    gather = "void madrona::mwhip::(anonymous namespace)::sortGather(args)"
    stats = "statsKernel(args)"
    _pmc_db(fetch, "FETCH_SIZE", {gather: [100.0] * 10, stats: [0.0] * 10})
    _pmc_db(write, "WRITE_SIZE", {gather: [300.0] * 15, stats: [0.0] * 15})
    sys.argv = ["make_traffic_json.py", "escape_room", "65536", fetch, write]
    m.main()
    entries = json.loads(capsys.readouterr().out)
    g = [e for e in entries if e["kernel"] == "SortArchetype:sort.gather"][0]
    assert g["fetch_size_kib_per_step"] == 100.0
    assert g["write_size_kib_per_step"] == 300.0       # not 300 * 15 / 10
    assert g["traffic_bytes"] == int((2 * 100.0 + 300.0) * 1024)
    assert g["replays_fetch_pass"] == 10 and g["replays_write_pass"] == 15
    step = m.step_entry("sim", 8, [
        {"kernel": "physics:worldStep(LDS)", "launches_per_step": 1.0, "traffic_bytes": 200},
        {"kernel": "__amd_rocclr_copyBuffer", "launches_per_step": 3.4,
         "traffic_bytes": 10 ** 9},
        {"kernel": "SortArchetype:sort.histogram", "launches_per_step": 0.0,
         "traffic_bytes": 7}])
    assert step["traffic_bytes"] == 200


def test_newest_committed_driver_line_schema():
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[2-9]_bench_default.json")))
    assert lines
    d = json.load(open(lines[-1]))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "steps/s" and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert "configs[2]" in d["config"]["workload"] and "model" not in d["config"]
    assert d["value"] == pytest.approx(
        d["config"]["total_worlds"] / (d["ms_per_step"] * 1e-3), rel=1e-6)
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "nodes"):
        assert key in r, key
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], abs=2e-4)
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and "sample" in c


def test_driver_line_stays_small_whatever_the_kernel_tables_hold():
    """BENCH_r04.json: `parsed: null` -- the one line had grown to 21 KB of
    per-kernel tables.  bench.driver_line keeps the contract's keys + five-number
    node summaries and sends the tables to a side file; here it is fed round 4's
    full record blown up to a 200-entry kernel table per section."""
    bench = _load("bench", os.path.join(ROOT, "bench.py"))
    full = json.loads(open(os.path.join(ROOT, "profiles", "r04_bench_default.json"))
                      .read().strip().splitlines()[-1])
    assert len(json.dumps(full)) > 20000
    fat = [{"name": f"sim::system{i}" + "x" * 60, "avg_us": 4.0 + i, "rows": 1e5,
            "algo_MB": 1.0, "GBps": 100.0, "bytes": "declared read/write set"}
           for i in range(200)]
    full["kernels"] = fat
    full["roofline"]["nodes"]["parallel_for"]["kernels"] = fat
    full["ecs_config2"]["kernels"] = fat
    full["render_config5"]["kernels"] = fat
    full["roofline"]["nodes"]["sort_node"]["chains"] *= 50
    line = bench.driver_line(full, "gpurun_out/bench_detail.json")
    text = json.dumps(line)
    assert len(text) < 8192 and len(text) < bench.LINE_BUDGET_BYTES
    assert "\n" not in text
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in line, key
    r = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["frac"] == full["roofline"]["frac"] and r["bound"] == "hbm"
    for node in ("sort_node", "parallel_for", "sort_and_parallel_for",
                 "physics_step_issue"):
        assert set(("achieved", "frac", "avg_us", "traffic", "algo_bytes")) <= set(
            r["nodes"][node]), node
    assert line["cpu_baseline"]["kind"] == "reference" and line["cpu_baseline"]["cores"] == 256
    assert line["value"] == full["value"] and line["detail"].endswith("bench_detail.json")
    assert line["ecs_config2"]["value"] == full["ecs_config2"]["value"]
    assert line["portable_sim"]["config3"]["value"] == full["portable_sim"]["config3"]["value"]


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` without torchrun (how the driver runs a scaling
    point) starts two ranks that rendezvous on 127.0.0.1; --dry-launch stops after
    the meeting (gloo, no GPU).  One workload for every N: the default simulator
    does not change with --gpus."""
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    lines = {}
    for n in (1, 2):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus",
                              str(n), "--dry-launch"], env=env, capture_output=True,
                             text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        printed = [l for l in out.stdout.splitlines() if l.strip()]
        assert len(printed) == 1, out.stdout       # ONE line, rank 0's
        lines[n] = json.loads(printed[0])
    two = lines[2]
    assert two["dry_launch"] and two["ranks"] == 2 and two["n_gpus"] == 2
    assert two["world_ranges"] == [[0, 8192], [8192, 8192]]
    assert two["total_worlds"] == 16384
    assert lines[1]["sim"] == two["sim"] == "escape_room_phys"


def test_driver_line_carries_every_baseline_config():
    """configs[0] (Cartpole) and configs[3]'s one-GPU share (Hide-and-Seek) ride on
    the driver line as one-number summaries next to configs[1], [2] and [4]."""
    bench = _load("bench", os.path.join(ROOT, "bench.py"))
    full = json.loads(open(os.path.join(ROOT, "profiles", "r04_bench_default.json"))
                      .read().strip().splitlines()[-1])
    full["hideseek_config4_share"] = {
        "value": 6.1e6, "ms_per_step": 1.34, "roofline": {"avg_us": 700.0},
        "kernels": [{"name": "x" * 80, "avg_us": 1.0}] * 100}
    full["cartpole_config1"] = {"value": 3.0e6, "ms_per_step": 0.02,
                                "cpu_reference": {"value": 1.0e6, "cores": 8}}
    full["config"]["rccl_ranks"] = 0
    line = bench.driver_line(full, "gpurun_out/bench_detail.json")
    assert line["hideseek_config4_share"] == {
        "value": 6.1e6, "ms_per_step": 1.34, "worlds": 8192, "physics_step_us": 700.0}
    assert line["cartpole_config1"]["cpu_reference"] == 1.0e6
    assert line["config"]["rccl_ranks"] == 0
    assert len(json.dumps(line)) < bench.LINE_BUDGET_BYTES
