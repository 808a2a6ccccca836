#!/usr/bin/env python3
"""Step / kernel timings of a bench workload under alternative builds and
switches, one process per variant, in ONE gpurun call (boxes differ by 8-25 %:
only numbers from the same call compare).

    python profiles/tools/run_variants.py SPEC.json [REPEATS] >> out.jsonl

SPEC.json: [{"label": "...", "sim": "escape_room_phys", "worlds": 8192,
             "build": "_build" | "_variants/<name>", "env": {"VAR": "1"},
             "kernels": "worldStep|lidar"}, ...]
(builds are cross-compiled beforehand: make -C madrona_amd OUT=_variants/<name>
EXTRA=-D... runtime _variants/<name>/lib<sim>_hip.so)"""
import json
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHILD = r"""
import json, re, sys
sys.path.insert(0, %r)
import bench, torch
torch.cuda.set_device(0)
sim, worlds, pat = sys.argv[1], int(sys.argv[2]), sys.argv[3]
r = bench.run_single(sim, worlds, 0, 5, 200, 300, 50, 30, settle=400)
ks = [(k["name"], k["avg_us"]) for k in r["kernels"] if re.search(pat, k["name"])]
print(json.dumps({"sim": sim, "worlds": worlds, "ms_per_step": r["ms_per_step"],
                  "steps_per_s": r["value"], "kernels": ks}))
""" % REPO

if __name__ == "__main__":
    spec = json.load(open(sys.argv[1]))
    repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    for rep in range(repeats):
        for v in spec:
            e = dict(os.environ)
            e["MADRONA_HIP_BUILD_DIR"] = v.get("build", "_build")
            e.update(v.get("env", {}))
            # (a variant that hangs must not take the GPU call with it: round 5
            # lost 25 GPU-minutes to ROC_SYSTEM_SCOPE_SIGNAL=0)
            try:
                out = subprocess.run([sys.executable, "-c", CHILD, v["sim"],
                                      str(v["worlds"]), v.get("kernels", "worldStep")],
                                     env=e, capture_output=True, text=True,
                                     timeout=float(v.get("timeout", 240)))
                line = [l for l in out.stdout.splitlines() if l.startswith("{")]
                rec = (json.loads(line[-1]) if line else
                       {"error": out.stderr[-600:]})
            except subprocess.TimeoutExpired:
                rec = {"error": "timed out"}
            rec.update(label=v["label"], build=v.get("build", "_build"),
                       env=v.get("env", {}), repeat=rep)
            print(json.dumps(rec), flush=True)
