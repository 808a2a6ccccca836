"""Device event log of MADRONA_TRACING builds (reference
src/mw/device/include/madrona/mw_gpu/tracing.hpp, src/mw/cuda_exec.cpp:204-257,
scripts/parse_device_tracing.py).

CPU: the parser on a synthetic log in the reference's record layout.
GPU: the Escape Room built with tracing runs six steps; the file the executor
writes on destruction holds, per step, a calibration record, one nodeStart /
nodeFinish pair per kernel of the graph in launch order, a blockStart /
blockWait pair per workgroup, and the results are those of the untraced build."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "madrona_amd", "scripts"))
import parse_device_tracing as pdt  # noqa: E402


def _rec(event, func, inv, node, block, sm, index, t):
    return (event, func, inv, node, 0, block, sm, index, t)


def test_parser_on_a_synthetic_log(tmp_path):
    recs = []
    for step in range(2):
        base = 1000000 * (step + 1)
        i = 0
        recs.append(_rec(pdt.CALIBRATION, 4, 0, 256, 0, 0, i, base)); i += 1
        for node, (wgs, dur) in enumerate([(3, 400), (1, 50)]):
            start = base + 1000 * (node + 1)
            recs.append(_rec(pdt.NODE_START, node, wgs * 256, node, 0, 0, i, start)); i += 1
            for wg in range(wgs):
                recs.append(_rec(pdt.BLOCK_START, node, wg, node, wg, wg % 2, i,
                                 start + 10 + wg)); i += 1
            for wg in range(wgs):
                recs.append(_rec(pdt.BLOCK_WAIT, node, wg, node, wg, wg % 2, i,
                                 start + 10 + dur + 10 * wg)); i += 1
        recs.append(_rec(pdt.BLOCK_EXIT, 0, 0, 2, 0, 0, i, base + 5000)); i += 1
        for node, (wgs, dur) in enumerate([(3, 400), (1, 50)]):
            start = base + 1000 * (node + 1)
            recs.append(_rec(pdt.NODE_FINISH, node, wgs * 256, node, wgs - 1, 0, i,
                             start + 10 + dur + 10 * (wgs - 1))); i += 1
    log = np.array(recs, dtype=pdt.RECORD)
    path = tmp_path / "t_madrona_device_tracing.bin"
    log.tofile(path)
    (tmp_path / "t_madrona_device_tracing_nodes.bin").write_text("alpha:pfor\nbeta:pfor\n")

    steps = pdt.split_steps(pdt.read_log(str(path)))
    assert len(steps) == 2
    r = pdt.analyse_step(steps[1], ["alpha:pfor", "beta:pfor"])
    assert r["total_ns"] == 5000 and r["compute_units"] == 256
    a, b = r["nodes"]
    assert (a["name"], a["workgroups"], a["compute_units"]) == ("alpha:pfor", 3, 2)
    assert a["start_ns"] == 1010 and a["duration_ns"] == 420
    assert a["workgroup_ns_max"] == 418 and b["duration_ns"] == 50
    assert abs(a["percent_of_kernels"] + b["percent_of_kernels"] - 100.0) < 1e-9
    assert pdt.main([str(path)]) == 0
    assert pdt.main([str(path), "--json", "--step", "0"]) == 0


CHILD = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, %r)
from madrona_amd.simlib import Simulator, hip_lib_path
worlds, steps = 256, 6
rng = np.random.default_rng(3)
with Simulator(hip_lib_path("escape_room"), worlds, seed=11, flags=8) as s:
    for _ in range(steps):
        a = np.stack([rng.integers(0, 4, (worlds, 2)), rng.integers(0, 8, (worlds, 2)),
                      rng.integers(-2, 3, (worlds, 2)), rng.integers(0, 2, (worlds, 2))],
                     -1).astype(np.int32)
        s.write_tensor("action", a)
        s.step(1)
    d = s.dump_all()
np.savez(sys.argv[1], **{k.replace(".", "__"): v[0] for k, v in d.items()})
""" % REPO


def _run(build_dir, out, env_extra):
    env = dict(os.environ)
    env["MADRONA_HIP_BUILD_DIR"] = build_dir
    env.update(env_extra)
    res = subprocess.run([sys.executable, "-c", CHILD, out], env=env,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]


@pytest.mark.gpu
def test_escape_room_device_trace(built, tmp_path):
    traced_lib = os.path.join(REPO, "madrona_amd", "_build_tracing", "libescape_room_hip.so")
    if not os.path.exists(traced_lib):
        pytest.skip("no MADRONA_TRACING build on this box")
    plain, traced = str(tmp_path / "plain.npz"), str(tmp_path / "traced.npz")
    _run("_build", plain, {})
    _run("_build_tracing", traced, {"MADRONA_MWHIP_TRACE_DIR": str(tmp_path),
                                    "MADRONA_MWGPU_TRACE_NAME": "esc"})
    # tracing changes nothing but the time
    a, b = np.load(plain), np.load(traced)
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k

    log_path = tmp_path / "esc_madrona_device_tracing.bin"
    names = (tmp_path / "esc_madrona_device_tracing_nodes.bin").read_text().splitlines()
    log = pdt.read_log(str(log_path))
    steps = pdt.split_steps(log)
    assert len(steps) == 6
    assert any("movementSystem" in n for n in names) and "stats:health" in names
    assert any(n.startswith("SortArchetype") or ":sort." in n for n in names)
    for step in steps:
        calib = step[0]
        assert calib["event"] == pdt.CALIBRATION and calib["funcID"] == 4
        assert calib["nodeID"] >= 64                      # compute units
        assert np.array_equal(np.sort(step["logIndex"]), np.arange(len(step)))
        r = pdt.analyse_step(step, names)
        nodes = r["nodes"]
        assert [n["nodeID"] for n in nodes] == list(range(len(nodes))) and len(nodes) >= 10
        starts = step[step["event"] == pdt.NODE_START]
        finishes = step[step["event"] == pdt.NODE_FINISH]
        assert len(starts) == len(finishes) == len(nodes)
        bs = step[step["event"] == pdt.BLOCK_START]
        bw = step[step["event"] == pdt.BLOCK_WAIT]
        assert len(bs) == len(bw) > len(nodes)
        prev_end = 0
        for n in nodes:
            assert n["workgroups"] >= 1, n
            assert n["duration_ns"] >= 0
            # kernels of a graph run one after the other (10 ns clock)
            assert n["marked_ns"] >= prev_end - 20, (n, prev_end)
            assert n["start_ns"] >= n["marked_ns"] - 20, n
            prev_end = n["start_ns"] + n["duration_ns"]
        assert r["total_ns"] >= prev_end - 20
        # per-row systems over 256 worlds x 2 agents: one workgroup; the
        # 256-thread ones report threads launched
        by_name = {n["name"]: n for n in nodes}
        mv = next(v for k, v in by_name.items() if "movementSystem" in k)
        assert mv["threads"] >= 512 and mv["compute_units"] >= 1
    out = subprocess.run([sys.executable,
                          os.path.join(REPO, "madrona_amd", "scripts", "parse_device_tracing.py"),
                          str(log_path), "--json"], capture_output=True, text=True)
    assert out.returncode == 0 and json.loads(out.stdout)["steps_in_log"] == 6
