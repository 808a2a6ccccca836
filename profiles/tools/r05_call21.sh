#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout 200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_final2.log 2>&1
tail -6 $O/pytest_gpu_final2.log
