#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke_final.log 2>&1
tail -2 $O/smoke_final.log
timeout 200 python profiles/tools/run_variants.py profiles/variants/r05_group_64k.json 2 > $O/group_64k_variants.jsonl 2> $O/group_64k_variants.err
python - <<PY
import json
for l in open("$O/group_64k_variants.jsonl"):
    d=json.loads(l); print(d.get("sim"), d.get("label","")[:50], round(d.get("ms_per_step",0)*1000,1), d.get("error","")[:300])
PY
