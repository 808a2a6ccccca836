#!/bin/bash
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
                  "raycast_us": r.get("avg_us")}))
PY
}
for w in 16384 32768 65536 131072 262144; do
run "geometry in HBM, 128 registers, $w workgroups" MADRONA_MWHIP_RAYCAST_GEO_LDS=0 MADRONA_HIP_BUILD_DIR=_variants/ray4 MADRONA_MWHIP_RAYCAST_WGS=$w
run "geometry in HBM, 5 wavefronts per SIMD, $w workgroups" MADRONA_MWHIP_RAYCAST_GEO_LDS=0 MADRONA_HIP_BUILD_DIR=_variants/ray5 MADRONA_MWHIP_RAYCAST_WGS=$w
done
