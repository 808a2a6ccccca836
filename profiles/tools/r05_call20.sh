#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout 300 python profiles/tools/run_variants.py profiles/variants/r05_row_chain.json 2 > $O/row_chain_variants.jsonl 2> $O/row_chain_variants.err
python - <<PY
import json
for l in open("$O/row_chain_variants.jsonl"):
    d=json.loads(l); print(d.get("sim"), d.get("label","")[:50], round(d.get("ms_per_step",0)*1000,1), d.get("error","")[:300])
PY
timeout 200 python -m pytest tests/test_exec_config_gpu.py "tests/test_parity_gpu.py::test_shared_launches_are_the_same_step" "tests/test_parity_gpu.py::test_escape_room_lockstep" -m gpu -q > $O/pytest_gpu_chain.log 2>&1
tail -8 $O/pytest_gpu_chain.log
