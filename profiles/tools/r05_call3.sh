#!/bin/bash
# Round 5, after the prune + fallback + opt-in shared rays: the GPU suite, then
# bench lines (default + driver command).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
tail -15 $O/pytest_gpu.log
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
cp gpurun_out/bench_detail.json $O/bench_default_detail.json
python - <<PY
import json
d=json.load(open("$O/bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["avg_us"], d.get("ecs_config2",{}).get("ms_per_step"), d.get("portable_sim"))
det=json.load(open("$O/bench_default_detail.json"))
for k in det["kernels"]: print(k["name"], k["avg_us"])
PY
python profiles/tools/run_variants.py profiles/variants/r05_eager.json 2 > $O/eager_variants.jsonl 2> $O/eager_variants.err
cat $O/eager_variants.jsonl | cut -c1-400
