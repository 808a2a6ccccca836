#!/bin/bash
# round 6, second GPU call: the whole GPU suite after the write-mask check,
# machine-scheduler strategies on the simulator's code object, the collective
# path of the bench on one rank
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_call2.log 2>&1
tail -5 $O/pytest_gpu_call2.log
timeout 600 python profiles/tools/run_variants.py profiles/variants/r06_phys_sched.json 2 > $O/r06_phys_sched.jsonl 2> $O/sched.err
timeout 300 python bench.py --gpus 1 --force-collective --no-secondary --no-cpu-baseline --steps 100 --warmup 20 > $O/bench_force_collective.json 2> $O/bench_force_collective.err
tail -c 600 $O/bench_force_collective.json
python - <<'PY'
import json
for l in open("gpurun_out/r06/r06_phys_sched.jsonl"):
    d = json.loads(l)
    print(d.get("repeat"), d.get("label", "")[:50], round(d.get("ms_per_step", 0), 4), d.get("kernels"), d.get("error", "")[-300:])
PY
