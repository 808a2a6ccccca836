#!/bin/bash
# A/B of builds / switches only: SPEC -> gpurun_out/r06/OUT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
timeout 1200 python profiles/tools/run_variants.py $SPEC ${REPEATS:-2} > $O/$OUT 2> $O/variants.err
python - <<PY
import json
for l in open("gpurun_out/r06/$OUT"):
    d = json.loads(l)
    print(d.get("repeat"), d.get("sim"), d.get("label", "")[:70], round(d.get("ms_per_step", 0), 4), d.get("kernels"), d.get("error", "")[-300:])
PY
