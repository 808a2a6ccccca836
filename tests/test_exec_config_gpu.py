"""GPU: MADRONA_MWHIP_EXEC_CONFIG_FILE (the reference's
MADRONA_MWGPU_EXEC_CONFIG_FILE, src/mw/cuda_exec.cpp:2115-2172: per task-graph
node the blocks per SM of the megakernel that runs it; here: the workgroups per
CU of the node's own kernel) and madrona_amd/scripts/profile.py, the tool that
fills it (reference scripts/profile.py)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from madrona_amd.simlib import Simulator, hip_lib_path
from parity_utils import compare_columns

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(worlds, steps):
    rng = np.random.default_rng(2)
    with Simulator(hip_lib_path("escape_room"), worlds, seed=9, flags=20) as s:
        for _ in range(steps):
            s.write_tensor("action", np.stack(
                [rng.integers(0, 4, (worlds, 2)), rng.integers(0, 8, (worlds, 2)),
                 rng.integers(-2, 3, (worlds, 2)), np.zeros((worlds, 2), int)],
                -1).astype(np.int32))
            s.step(1)
        stats = s.profile(3)
        return s.dump_all(), stats


def test_exec_config_caps_node_grids(built, tmp_path, monkeypatch):
    worlds = 3000
    monkeypatch.delenv("MADRONA_MWHIP_EXEC_CONFIG_FILE", raising=False)
    plain, base = _run(worlds, 25)
    # (a launch shared by several nodes carries its first node's index and the
    # largest of their grids)
    nodes = [k for k in base if k["node_index"] != 0xFFFFFFFF and k["rows"] > 0]
    assert len(nodes) >= 6 and any(k["name"].startswith("group[") for k in nodes)
    # every ParallelFor node on one workgroup per CU
    cfg = tmp_path / "node_config.json"
    cfg.write_text(json.dumps({str(k["node_index"]): 1 for k in nodes}))
    monkeypatch.setenv("MADRONA_MWHIP_EXEC_CONFIG_FILE", str(cfg))
    capped, stats = _run(worlds, 25)
    by_node = {k["node_index"]: k for k in stats}
    shrunk = 0
    for k in nodes:
        now = by_node[k["node_index"]]
        assert now["name"] == k["name"]
        assert now["workgroups"] <= max(256, 1) and now["workgroups"] <= k["workgroups"]
        shrunk += now["workgroups"] < k["workgroups"]
    assert shrunk >= 1
    # the grid decides when a row is visited, never the result
    assert not compare_columns(plain, capped)

    # a file that is not the reference's flat { "<node>": <count> } object: the
    # executor cannot be built, which -- as in the reference, FATAL at
    # cuda_exec.cpp:2146 -- ends the process (checked in a child)
    cfg.write_text('{"movement": 3}')
    child = subprocess.run(
        [sys.executable, "-c",
         "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
         "from madrona_amd.simlib import Simulator, hip_lib_path\n"
         "Simulator(hip_lib_path('escape_room'), 64, seed=1, flags=20)\n"
         "print('built')" % (REPO, os.path.join(REPO, "tests"))],
        capture_output=True, text=True, timeout=300)
    assert child.returncode != 0 and "built" not in child.stdout
    assert "EXEC_CONFIG_FILE" in child.stderr


def test_profile_script_writes_a_config(built, tmp_path, monkeypatch):
    monkeypatch.delenv("MADRONA_MWHIP_EXEC_CONFIG_FILE", raising=False)
    out = tmp_path / "node_config.json"
    r = subprocess.run([sys.executable,
                        os.path.join(REPO, "madrona_amd", "scripts", "profile.py"),
                        "--sim", "escape_room", "--worlds", "2048", "--steps", "40",
                        "--reps", "10", "--candidates", "1,4", "--out", str(out)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-800:]
    config = json.loads(out.read_text())
    report = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(report) >= 8 and all("default_us" in e for e in report)
    assert all(k.isdigit() and v in (1, 4) for k, v in config.items())
    # and the executor takes it
    monkeypatch.setenv("MADRONA_MWHIP_EXEC_CONFIG_FILE", str(out))
    _run(256, 3)


def test_declared_read_write_sets_reach_the_profile(built, monkeypatch):
    """madrona::mwhip::systemIO ("each node declares its read/write
    set next to the kernel"): a declared system is priced at rows x (4 + reads +
    writes), flagged io_declared; the numbers are the component sizes of
    sims/escape_room/sim.hpp."""
    monkeypatch.delenv("MADRONA_MWHIP_EXEC_CONFIG_FILE", raising=False)
    worlds = 512
    # a grouped launch (nodes that named the same dependencies share one) is
    # priced at the sum over its nodes
    with Simulator(hip_lib_path("escape_room"), worlds, seed=1, flags=0) as s:
        s.step(5)
        grouped = {k["name"]: k for k in s.profile(4)}
    group = [k for n, k in grouped.items() if n.startswith("group[")
             and "escape::rewardSystem" in n and "escape::stepTrackerSystem" in n
             and "escape::buttonSystem" in n]
    assert len(group) == 1 and group[0]["io_declared"]
    # buttons 6 x (4 + Position 12 + 2 x Position 12 + ButtonState 4) + agents 2 x
    # (28 + 16) per world
    assert group[0]["rows"] == (6 + 2 + 2) * worlds
    assert group[0]["algo_bytes"] == pytest.approx(
        (6 * 44 + 2 * 28 + 2 * 16) * worlds)
    # ... node by node with the grouping off
    monkeypatch.setenv("MADRONA_MWHIP_GROUP", "0")
    with Simulator(hip_lib_path("escape_room"), worlds, seed=1, flags=0) as s:
        s.step(5)
        stats = {k["name"]: k for k in s.profile(4)}
    per_row = {
        # WorldID 4 + Action 16 + Rotation 16 -> ExternalForce 12 + ExternalTorque 12
        "escape::movementSystem": (60, 2 * worlds),
        # Position 12 + Progress 4 -> Progress 4 + Reward 4
        "escape::rewardSystem": (28, 2 * worlds),
        # StepsRemaining 4 -> StepsRemaining 4 + Done 4
        "escape::stepTrackerSystem": (16, 2 * worlds),
        # the two door systems as one node (madrona::mwhip::rowChain): DoorProperties
        # 40 + 2 x ButtonState 4 + OpenState 4 + Position 12 -> OpenState 4 + Position 12
        "chain[escape::doorOpenSystem > escape::setDoorPositionSystem]": (84, 3 * worlds),
    }
    for name, (bytes_per_row, rows) in per_row.items():
        k = stats[name]
        assert k["io_declared"], name
        assert k["rows"] == rows, (name, k["rows"])
        assert k["algo_bytes"] == pytest.approx(bytes_per_row * rows), name
    # every system of the simulator is declared; the runtime's own kernels are not
    systems = [k for n, k in stats.items() if "escape::" in n]
    assert len(systems) >= 9 and all(k["io_declared"] for k in systems)
    assert not stats["stats:health"]["io_declared"]
