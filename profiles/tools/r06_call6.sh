#!/bin/bash
# round 6: GPU suite on the dieted physics step, the BASELINE-size lock steps on
# its two-wavefronts-per-SIMD build, A/B
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_call6.log 2>&1
tail -5 $O/pytest_gpu_call6.log
MADRONA_HIP_BUILD_DIR=_variants/diet_2waves timeout 600 python -m pytest tests/test_parity_gpu.py -q -k "full_size_lockstep and (phys or hideseek)" > $O/pytest_gpu_call6_2waves.log 2>&1
tail -5 $O/pytest_gpu_call6_2waves.log
timeout 900 python profiles/tools/run_variants.py profiles/variants/r06_diet2.json 2 > $O/r06_diet2_variants.jsonl 2> $O/variants.err
python - <<PY
import json
for l in open("gpurun_out/r06/r06_diet2_variants.jsonl"):
    d = json.loads(l)
    print(d.get("repeat"), d.get("sim"), d.get("label", "")[:60], round(d.get("ms_per_step", 0), 4), d.get("kernels"), d.get("error", "")[-300:])
PY
