#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh
mkdir -p $O
cd $R
bash $R/profiles/tools/refresh_r06.sh > $O/refresh.log 2>&1
tail -3 $O/refresh.log
head -c 900 $O/r06_bench_default.json
