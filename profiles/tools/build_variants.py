#!/usr/bin/env python3
"""Step / kernel timings of a bench workload under alternative builds of its
simulator library (madrona_amd/_variants/<name>/, cross-compiled beforehand with
`make OUT=_variants/<name> EXTRA=-D...`), one process per build.

    python profiles/tools/build_variants.py SIM WORLDS KERNEL_SUBSTRING DIR [DIR ...]
"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHILD = r"""
import json, sys
sys.path.insert(0, %r)
import bench, torch
torch.cuda.set_device(0)
sim, worlds, pat = sys.argv[1], int(sys.argv[2]), sys.argv[3]
r = bench.run_single(sim, worlds, 0, 5, 200, 200, 50, 30, settle=300)
ks = [(k["name"], k["avg_us"]) for k in r["kernels"] if pat in k["name"]]
print(json.dumps({"sim": sim, "worlds": worlds, "ms_per_step": r["ms_per_step"],
                  "kernels": ks}))
""" % REPO

if __name__ == "__main__":
    sim, worlds, pat = sys.argv[1], sys.argv[2], sys.argv[3]
    for d in sys.argv[4:]:
        e = dict(os.environ)
        e["MADRONA_HIP_BUILD_DIR"] = d
        out = subprocess.run([sys.executable, "-c", CHILD, sim, worlds, pat], env=e,
                             capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        rec = json.loads(line[-1]) if line else {"error": out.stderr[-400:]}
        rec["build"] = d
        print(json.dumps(rec), flush=True)
