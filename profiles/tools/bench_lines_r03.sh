#!/bin/bash
# The bench lines of round 3 alone (after profiles/rNN_hbm_traffic.json and
# rNN_issue_counters.json exist: bench.py copies them into its rooflines).
ROUND=${ROUND:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh
mkdir -p $O
timeout 300 python $R/bench.py > $O/${ROUND}_bench_default.json 2> $O/bench_default.err
timeout 200 python $R/bench.py --sim hideseek > $O/${ROUND}_bench_hideseek_w8192.json 2> $O/bench_hideseek.err
timeout 200 python $R/bench.py --sim escape_room --steps 3000 > $O/${ROUND}_bench_escape_room_w4096.json 2> $O/bench_er.err
timeout 200 python $R/bench.py --sim escape_room --worlds 65536 --steps 300 --no-cpu-baseline > $O/${ROUND}_bench_escape_room_w65536.json 2> $O/bench_er64k.err
timeout 300 python $R/bench.py --sim escape_room_render > $O/${ROUND}_bench_escape_room_render_w8192.json 2> $O/bench_render.err
