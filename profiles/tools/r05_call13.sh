#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
python profiles/tools/run_variants.py profiles/variants/r05_preload.json 2 > $O/preload_variants.jsonl 2> $O/preload_variants.err
python - <<PY
import json
for l in open("$O/preload_variants.jsonl"):
    d=json.loads(l)
    print(d.get("sim"), d.get("label","")[:40], round(d.get("ms_per_step",0),4), [(k[0].split("::")[-1][:14], k[1]) for k in d.get("kernels",[])], d.get("error","")[:200])
PY
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_preload.log 2>&1
tail -6 $O/pytest_gpu_preload.log
