#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
timeout 300 python profiles/tools/phys_phase_cycles.py 8192 > $O/${PHASES:-r06_physics_phases_diet_b.txt} 2>&1
grep -v amdgpu $O/${PHASES:-r06_physics_phases_diet_b.txt} | head -14
timeout 900 python profiles/tools/run_variants.py ${SPEC:-profiles/variants/r06_diet3.json} 2 > $O/${OUT:-r06_diet3_variants.jsonl} 2> $O/variants.err
python - <<PY
import json
for l in open("gpurun_out/r06/${OUT:-r06_diet3_variants.jsonl}"):
    d = json.loads(l)
    print(d.get("repeat"), d.get("sim"), d.get("label", "")[:60], round(d.get("ms_per_step", 0), 4), d.get("kernels"), d.get("error", "")[-300:])
PY
