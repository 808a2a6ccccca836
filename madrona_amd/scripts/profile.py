#!/usr/bin/env python3
"""Profile-guided per-node launch configuration (the PGO half).

The reference: scripts/profile.py runs a simulator under a sweep of megakernel
configurations (blocks per SM), reads the per-node times out of the device
trace and writes, per task-graph node, the configuration it ran fastest with to
a JSON file that MWCudaExecutor reads through MADRONA_MWGPU_EXEC_CONFIG_FILE
(src/mw/cuda_exec.cpp:2115-2172: { "<node index>": <blocks per SM> }).

Here a node is its own kernel, and what the executor can choose per node at
graph-build time is how many workgroups per CU the kernel occupies (its grid:
ParallelFor kernels stride over their rows).  This script

  1. steps the simulator once per candidate (MADRONA_MWHIP_EXEC_CONFIG_FILE set
     to a file that gives EVERY node that candidate), taking per-node times
     from the executor's own profile (HIP events on the dispatches,
     mwhip_profile -- the numbers the device trace of madrona/mw_gpu/tracing.hpp
     gives, without a tracing build);
  2. keeps, per node, the candidate with the lowest time if it beats the
     default launch by more than --min-gain;
  3. writes { "<node index>": <workgroups per CU> } -- the reference's format --
     to --out; export MADRONA_MWHIP_EXEC_CONFIG_FILE=<that file> to use it.

Compile-time choices (madrona::mwhip::systemWavesPerSIMD, the register budget of
a system's kernel) are swept with profiles/tools/run_variants.py over builds in
madrona_amd/_variants/; this script covers what needs no rebuild.

    python madrona_amd/scripts/profile.py --sim escape_room --worlds 4096 \\
        --out node_config.json
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHILD = r"""
import json, sys
sys.path.insert(0, %r)
import numpy as np
import torch
from madrona_amd.simlib import Simulator, hip_lib_path
sim, worlds, steps, reps, flags = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), \
    int(sys.argv[4]), int(sys.argv[5])
with Simulator(hip_lib_path(sim), worlds, seed=5, flags=flags) as s:
    if "action" in s.tensor_names:
        _, dtype, dims, _ = s._tensor_info["action"]
        rng = np.random.default_rng(0)
        s.write_tensor("action", rng.integers(0, 2, dims).astype(dtype))
    s.step(steps)
    stats = s.profile(reps)
print(json.dumps([{"name": k["name"], "node": k["node_index"], "us": k["avg_us"],
                   "workgroups": k["workgroups"], "rows": k["rows"]}
                  for k in stats]))
""" % REPO


def measure(args, per_cu):
    """Per-kernel stats with every node capped at `per_cu` workgroups per CU
    (None: the executor's default launch)."""
    env = dict(os.environ)
    env.pop("MADRONA_MWHIP_EXEC_CONFIG_FILE", None)
    # every node timed in its own launch (the step itself puts nodes that named
    # the same dependencies into one, sized for the largest of their grids)
    env["MADRONA_MWHIP_GROUP"] = "0"
    tmp = None
    if per_cu is not None:
        tmp = tempfile.NamedTemporaryFile("w", suffix=".json", delete=False)
        json.dump({str(i): per_cu for i in range(args.max_nodes)}, tmp)
        tmp.close()
        env["MADRONA_MWHIP_EXEC_CONFIG_FILE"] = tmp.name
    out = subprocess.run([sys.executable, "-c", CHILD, args.sim, str(args.worlds),
                          str(args.steps), str(args.reps), str(args.flags)],
                         env=env, capture_output=True, text=True)
    if tmp is not None:
        os.unlink(tmp.name)
    lines = [l for l in out.stdout.splitlines() if l.startswith("[")]
    if not lines:
        raise RuntimeError(out.stderr[-800:])
    return json.loads(lines[-1])


def main():
    p = argparse.ArgumentParser(description=__doc__,
                                formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--sim", default="escape_room")
    p.add_argument("--worlds", type=int, default=4096)
    p.add_argument("--flags", type=int, default=200, help="simulator flags (auto-reset 1/N)")
    p.add_argument("--steps", type=int, default=300, help="steps before measuring")
    p.add_argument("--reps", type=int, default=30, help="profiled steps per candidate")
    p.add_argument("--candidates", default="1,2,4,8",
                   help="workgroups per CU to try for every node")
    p.add_argument("--min-gain", type=float, default=0.03,
                   help="keep a candidate only if it beats the default by this fraction")
    p.add_argument("--max-nodes", type=int, default=256)
    p.add_argument("--out", default="node_config.json")
    args = p.parse_args()

    base = measure(args, None)
    runs = {c: measure(args, c) for c in [int(c) for c in args.candidates.split(",")]}

    config, report = {}, []
    for i, k in enumerate(base):
        if k["node"] == 0xFFFFFFFF:
            continue            # not a node's own kernel (sort chains, health ...)
        best_c, best_us = None, k["us"]
        for c, stats in runs.items():
            if i < len(stats) and stats[i]["name"] == k["name"] and \
                    stats[i]["workgroups"] != k["workgroups"] and \
                    stats[i]["us"] < best_us:
                best_c, best_us = c, stats[i]["us"]
        kept = best_c is not None and best_us < k["us"] * (1.0 - args.min_gain)
        if kept:
            config[str(k["node"])] = best_c
        report.append({"node": k["node"], "name": k["name"], "default_us": round(k["us"], 2),
                       "default_workgroups": k["workgroups"],
                       "best_workgroups_per_cu": best_c if kept else None,
                       "best_us": round(best_us, 2) if kept else None})
    with open(args.out, "w") as f:
        json.dump(config, f, indent=1)
    for r in report:
        print(json.dumps(r))
    print(f"wrote {args.out}: {len(config)} of {len(report)} nodes tuned",
          file=sys.stderr)


if __name__ == "__main__":
    main()
