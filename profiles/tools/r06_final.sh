#!/bin/bash
# the round's last GPU call: the whole GPU suite, then the refresh of profiles/
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q > $O/r06_pytest_gpu_final.log 2>&1
tail -5 $O/r06_pytest_gpu_final.log
bash $R/profiles/tools/refresh_r06.sh > $O/refresh.log 2>&1
tail -3 $O/refresh.log
head -c 600 $O/r06_bench_default.json
