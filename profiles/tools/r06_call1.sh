#!/bin/bash
# round 6, first GPU call: what a register cap costs the physics step by itself,
# then the driver's own bench command (new summaries on the line)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
timeout 900 python profiles/tools/run_variants.py profiles/variants/r06_phys_caps.json 2 > $O/r06_phys_caps.jsonl 2> $O/caps.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
tail -c 1500 $O/bench_default.json
python - <<'PY'
import json
for l in open("gpurun_out/r06/r06_phys_caps.jsonl"):
    d = json.loads(l)
    print(d.get("repeat"), d.get("label", "")[:60], d.get("ms_per_step"), d.get("kernels"), d.get("error", "")[-300:])
PY
