#!/bin/bash
# round 6: GPU suite + A/B of the candidate pass of the physics step
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_call4.log 2>&1
tail -5 $O/pytest_gpu_call4.log
SPEC=${SPEC:-profiles/variants/r06_cand.json}
OUT=${OUT:-r06_cand_variants.jsonl}
timeout 900 python profiles/tools/run_variants.py $SPEC 2 > $O/$OUT 2> $O/variants.err
python - <<PY
import json
for l in open("gpurun_out/r06/$OUT"):
    d = json.loads(l)
    print(d.get("repeat"), d.get("sim"), d.get("label", "")[:60], round(d.get("ms_per_step", 0), 4), d.get("kernels"), d.get("error", "")[-300:])
PY
