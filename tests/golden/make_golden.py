"""Generates the golden fixtures in this directory from the REFERENCE CPU
backend (oracle/_ref, compiled from /root/reference by oracle/Makefile).
Run here (the reference tree does not exist on the GPU box):

    make -C oracle parity && python tests/golden/make_golden.py

Each fixture stores, for a few checkpoints, every dumped column (raw bytes) and
rows-per-world of a seeded run; escape_room additionally stores the action
tensors that were fed in, so the HIP run can replay them.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from madrona_amd.simlib import Simulator, ref_lib_path  # noqa: E402

CASES = {
    # name: (sim, worlds, seed, flags, checkpoints)
    "cartpole_w64": ("cartpole", 64, 5, 0, [1, 10, 100, 400]),
    "escape_room_w16": ("escape_room", 16, 5, 20, [1, 5, 25, 60]),
    "sort_stress_w33": ("sort_stress", 33, 7, 0, [1, 3, 10, 40]),
    # rigid-body physics on: BVH + narrowphase + XPBD + joints (grab action)
    "escape_room_phys_w8": ("escape_room_phys", 8, 5, 30, [1, 5, 25, 60]),
    # Hide-and-Seek shape: 29 bodies/world, wedge hulls, lock/unlock, ray casts
    "hideseek_w8": ("hideseek", 8, 5, 40, [1, 5, 25, 110]),
    # sphere primitives (GJK sphere-hull), random kicks, no actions
    "ball_pit_w8": ("ball_pit", 8, 5, 50, [1, 10, 40, 120]),
}

AGENTS = {"escape_room": 2, "escape_room_phys": 2, "hideseek": 5}


def escape_actions(step, num_worlds, grab=False, agents=2):
    rng = np.random.default_rng(1000 + step)
    shape = (num_worlds, agents)
    return np.stack([
        rng.integers(0, 4, shape), rng.integers(0, 8, shape),
        rng.integers(-2, 3, shape),
        rng.integers(0, 2, shape) if grab else np.zeros(shape, int),
    ], -1).astype(np.int32)


def actions_for(sim, step, num_worlds):
    """The action tensor fed at `step`, or None for simulators without one
    (move amount, move angle, turn, grab / lock)."""
    if sim not in AGENTS:
        return None
    return escape_actions(step, num_worlds, grab=sim != "escape_room",
                          agents=AGENTS[sim])


def main():
    only = sys.argv[1:]
    for name, (sim, worlds, seed, flags, checkpoints) in CASES.items():
        if only and name not in only:
            continue
        out = {}
        with Simulator(ref_lib_path(sim), worlds, seed=seed, num_workers=1,
                       flags=flags) as s:
            for step in range(1, max(checkpoints) + 1):
                actions = actions_for(sim, step, worlds)
                if actions is not None:
                    s.write_tensor("action", actions)
                s.step(1)
                if step in checkpoints:
                    for col, (rows, counts) in s.dump_all().items():
                        out[f"s{step}/{col}/rows"] = rows
                        out[f"s{step}/{col}/counts"] = counts
        out["meta"] = np.array([worlds, seed, flags] + checkpoints, dtype=np.int64)
        path = os.path.join(HERE, f"{name}.npz")
        np.savez_compressed(path, **out)
        print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


if __name__ == "__main__":
    main()
