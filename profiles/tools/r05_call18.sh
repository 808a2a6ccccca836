#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout 900 python profiles/tools/run_variants.py profiles/variants/r05_group_vgprs.json 2 > $O/group_vgprs_variants.jsonl 2> $O/group_vgprs_variants.err
python - <<PY
import json
for l in open("$O/group_vgprs_variants.jsonl"):
    d=json.loads(l); print(d.get("sim"), d.get("label","")[:50], round(d.get("ms_per_step",0)*1000,1), d.get("error","")[:300])
PY
timeout 600 python -m pytest tests/test_exec_config_gpu.py -m gpu -q > $O/pytest_gpu_group2.log 2>&1
tail -8 $O/pytest_gpu_group2.log
