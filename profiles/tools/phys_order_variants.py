#!/usr/bin/env python3
"""Config 3 with and without the heaviest-worlds-first order of the physics step
(MADRONA_MWHIP_PHYS_ORDER), one process each: step time + the kernels involved."""
import json, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r"""
import json, sys
sys.path.insert(0, %r)
import bench, torch
torch.cuda.set_device(0)
sim = sys.argv[1]
r = bench.run_single(sim, 8192, 0, 5, 200, 200, 50, 30, settle=300)
print(json.dumps({"sim": sim, "ms_per_step": r["ms_per_step"], "kernels": [(k["name"], k["avg_us"]) for k in r["kernels"]]}))
""" % REPO
for sim in ("escape_room_phys", "hideseek"):
    for name, env in [("index order", {"MADRONA_MWHIP_PHYS_ORDER": "0"}), ("heaviest first", {})]:
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, "-c", CHILD, sim], env=e, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        rec = json.loads(line[-1]) if line else {"error": out.stderr[-600:]}
        rec["variant"] = name
        print(json.dumps(rec), flush=True)
