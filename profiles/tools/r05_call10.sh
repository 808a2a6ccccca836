#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu_full.log 2>&1
tail -12 $O/pytest_gpu_full.log
python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -5 $O/smoke.log
