#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout 900 python profiles/tools/run_variants.py profiles/variants/r05_dag.json 2 > $O/dag_variants.jsonl 2> $O/dag_variants.err
python - <<PY
import json
for l in open("$O/dag_variants.jsonl"):
    d=json.loads(l); print(d.get("sim"), d.get("label","")[:34], round(d.get("ms_per_step",0)*1000,1), d.get("error","")[:300])
PY
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_dag.log 2>&1
tail -8 $O/pytest_gpu_dag.log
