#!/bin/bash
# GPU suite, then an A/B: SPEC -> gpurun_out/r06/OUT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_latest.log 2>&1
tail -4 $O/pytest_gpu_latest.log
bash profiles/tools/r06_ab.sh
