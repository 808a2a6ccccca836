#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -q -x -k "breadth_first_bvh" > $O/pytest_gpu_rebuild.log 2>&1
tail -12 $O/pytest_gpu_rebuild.log
python profiles/tools/run_variants.py profiles/variants/r05_rebuild.json 2 > $O/rebuild_variants.jsonl 2> $O/rebuild_variants.err
cut -c1-420 $O/rebuild_variants.jsonl
tail -3 $O/rebuild_variants.err
timeout 900 python -m pytest tests -m gpu -q -x -k "physics or ball_pit or hideseek or escape_room_phys or broadphase or rays or crowded or full_size or golden or tgs" > $O/pytest_gpu_rebuild2.log 2>&1
tail -6 $O/pytest_gpu_rebuild2.log
