#!/usr/bin/env python3
"""Sort-node timings of the bench workloads under the executor's sort switches
(one process per variant: the switches are read when an executor is created).

    python profiles/tools/sort_variants.py > gpurun_out/sort_variants.jsonl

Prints one JSON line per (workload, variant): the sort kernels of a step with
their event-timed averages, and the node's SURVEY 8d roofline fraction."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
WORKLOADS = [("escape_room_phys", 8192), ("escape_room", 4096), ("escape_room", 65536)]
VARIANTS = {
    "radix": {"MADRONA_MWHIP_SORT_COMPACT": "0", "MADRONA_MWHIP_GATHER_WIDE": "0"},
    "radix+wide": {"MADRONA_MWHIP_SORT_COMPACT": "0"},
    "compact": {"MADRONA_MWHIP_GATHER_WIDE": "0"},
    "compact+wide": {},
}

CHILD = r"""
import json, sys
sys.path.insert(0, %r)
import bench, torch
torch.cuda.set_device(0)
sim, worlds = sys.argv[1], int(sys.argv[2])
r = bench.run_single(sim, worlds, 0, 5, 200, 200, 50, 30, settle=300)
sort = [k for k in r["kernels"] if ":sort." in k["name"]]
node = (r["roofline"] or {}).get("nodes", {}).get("sort_node")
print(json.dumps({"sim": sim, "worlds": worlds, "ms_per_step": r["ms_per_step"],
                  "sort_kernels": sort, "sort_node_us": node and node["avg_us"],
                  "sort_node_frac": node and node["frac"],
                  "sort_node_MB": node and node["algo_bytes_per_launch"] / 1e6}))
""" % REPO

for sim, worlds in WORKLOADS:
    for name, env in VARIANTS.items():
        e = dict(os.environ)
        e.update(env)
        out = subprocess.run([sys.executable, "-c", CHILD, sim, str(worlds)], env=e,
                             capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(json.dumps({"sim": sim, "worlds": worlds, "variant": name,
                              "error": out.stderr[-400:]}), flush=True)
            continue
        rec = json.loads(line[-1])
        rec["variant"] = name
        print(json.dumps(rec), flush=True)
