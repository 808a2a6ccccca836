#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
python profiles/tools/run_variants.py profiles/variants/r05_runtime_env.json 1 > $O/runtime_env.jsonl 2> $O/runtime_env.err
python - <<PY
import json
for l in open("$O/runtime_env.jsonl"):
    d=json.loads(l); print(d.get("label"), round(d.get("ms_per_step",0)*1000,1), d.get("error","")[:150])
PY
