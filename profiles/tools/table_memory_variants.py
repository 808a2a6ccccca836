import json, os, subprocess, sys
REPO="/root/repo"
CHILD = r"""
import json, sys
sys.path.insert(0, %r)
import bench, torch
torch.cuda.set_device(0)
r = bench.run_single("escape_room_phys", 8192, 0, 5, 200, 200, 50, 30, settle=300)
print(json.dumps({"ms_per_step": r["ms_per_step"], "kernels": [(k["name"], k["avg_us"]) for k in r["kernels"]]}))
""" % REPO
for name, env in [("vmm (default)", {}), ("plain allocations", {"MADRONA_MWHIP_TABLE_GROWTH": "1"}), ("vmm (default)", {})]:
    e = dict(os.environ); e.update(env)
    out = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    rec = json.loads(line[-1]) if line else {"error": out.stderr[-600:]}
    rec["variant"] = name
    print(json.dumps(rec), flush=True)
