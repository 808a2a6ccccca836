#!/bin/bash
# ray caster occupancy variants (config 5's render pass), one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() {  # label, env...
  label=$1; shift
  env "$@" timeout 400 python bench.py --sim escape_room_render --no-cpu-baseline > /tmp/rc.json 2>/dev/null
  python - "$label" <<'PY'
import json, sys
d = json.load(open('/tmp/rc.json'))
r = d['roofline']
print(json.dumps({"label": sys.argv[1], "steps_per_s": d["value"], "ms_per_step": d["ms_per_step"],
                  "raycast_us": r.get("avg_us"), "kernel": r.get("kernel", "")[:40]}))
PY
}
for rep in 1 2; do
run "geometry in LDS, 3 wavefronts per SIMD (default)" A=1
run "geometry in HBM (24 KB LDS block), 149 registers: 3 wavefronts per SIMD" MADRONA_MWHIP_RAYCAST_GEO_LDS=0
run "geometry in HBM, capped at 128 registers: 4 wavefronts per SIMD" MADRONA_MWHIP_RAYCAST_GEO_LDS=0 MADRONA_HIP_BUILD_DIR=_variants/ray4
run "geometry in HBM, 128 registers, 8 workgroups per CU" MADRONA_MWHIP_RAYCAST_GEO_LDS=0 MADRONA_HIP_BUILD_DIR=_variants/ray4 MADRONA_MWHIP_RAYCAST_WGS=2048
done
