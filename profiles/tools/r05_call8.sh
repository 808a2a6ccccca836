#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
python profiles/tools/run_variants.py profiles/variants/r05_refresh2.json 2 > $O/refresh2_variants.jsonl 2> $O/refresh2_variants.err
cut -c1-500 $O/refresh2_variants.jsonl
timeout 900 python -m pytest tests -m gpu -q -x -k "physics or ball_pit or hideseek or escape_room_phys or broadphase or rays or crowded or full_size or golden or tgs" > $O/pytest_gpu_refresh2.log 2>&1
tail -8 $O/pytest_gpu_refresh2.log
