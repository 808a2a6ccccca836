#!/usr/bin/env python3
"""Sort-node timings of the bench workloads under the executor's sort switches
(one process per variant: the switches are read when an executor is created).

    python profiles/tools/sort_variants.py > gpurun_out/sort_variants.jsonl

Prints one JSON line per (workload, variant): the sort kernels of a step with
their event-timed averages, and the node's SURVEY 8d roofline fraction; first
line: the box's measured copy / triad bandwidth."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
RADIX = {"MADRONA_MWHIP_SORT_COMPACT": "0"}
APART = {"MADRONA_MWHIP_SORT_CARRIES_MISC": "0"}
RUNS = [
    ("escape_room_phys", 8192, "radix", RADIX),
    ("escape_room_phys", 8192, "compact", {}),
    ("escape_room_phys", 8192, "compact, misc ops in a launch of their own", APART),
    ("escape_room_phys", 8192, "compact, b4096", {"MADRONA_MWHIP_GATHER_BLOCKS": "4096"}),
    ("escape_room_phys", 8192, "compact, narrow gather", {"MADRONA_MWHIP_GATHER_WIDE": "0"}),
    ("escape_room", 4096, "radix", RADIX),
    ("escape_room", 4096, "compact", {}),
    ("escape_room", 65536, "radix", RADIX),
    ("escape_room", 65536, "compact", {}),
    ("escape_room", 65536, "compact, b8192", {"MADRONA_MWHIP_GATHER_BLOCKS": "8192"}),
]

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

if __name__ == "__main__":
    import bench
    import torch
    torch.cuda.set_device(0)
    print(json.dumps({"hbm": bench.measure_hbm_bandwidth()}), flush=True)
    for sim, worlds, name, env in RUNS:
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
