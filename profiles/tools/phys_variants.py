#!/usr/bin/env python3
"""Fused physics step under the executor's switches (one process per variant).

    python profiles/tools/phys_variants.py > gpurun_out/phys_variants.jsonl"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ONE = {"MADRONA_MWHIP_PHYS_LANES": "64"}
PACK = {"MADRONA_MWHIP_PHYS_PACK": "1"}
RUNS = [
    ("escape_room_phys", 8192, "one world per wavefront (2 waves/SIMD)", ONE),
    ("escape_room_phys", 8192, "two worlds per wavefront (1 wave/SIMD)", {}),
    ("escape_room_phys", 8192, "two worlds per wavefront + world images", PACK),
    ("escape_room_phys", 8192, "one world per wavefront + world images",
     {**ONE, **PACK}),
    ("hideseek", 8192, "one world per wavefront (2 waves/SIMD)", ONE),
    ("hideseek", 8192, "two worlds per wavefront (1 wave/SIMD)", {}),
    ("hideseek", 8192, "two worlds per wavefront + world images", PACK),
]

CHILD = r"""
import json, sys
sys.path.insert(0, %r)
import bench, torch
torch.cuda.set_device(0)
sim, worlds = sys.argv[1], int(sys.argv[2])
r = bench.run_single(sim, worlds, 0, 5, 200, 300, 50, 30, settle=400)
phys = [k for k in r["kernels"] if "worldStep" in k["name"] or "packWorlds" in k["name"]]
print(json.dumps({"sim": sim, "worlds": worlds, "ms_per_step": r["ms_per_step"],
                  "value": r["value"], "physics_step_us": [k["avg_us"] for k in phys]}))
""" % REPO

if __name__ == "__main__":
    for sim, worlds, name, env in RUNS:
        e = dict(os.environ)
        e.update(env)
        out = subprocess.run([sys.executable, "-c", CHILD, sim, str(worlds)], env=e,
                             capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(json.dumps({"sim": sim, "worlds": worlds, "variant": name,
                              "error": out.stderr[-600:]}), flush=True)
            continue
        rec = json.loads(line[-1])
        rec["variant"] = name
        print(json.dumps(rec), flush=True)
