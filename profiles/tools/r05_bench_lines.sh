#!/bin/bash
# the bench lines of refresh_r05.sh alone (with their detail files)
ROUND=${ROUND:-r05}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py > $O/${ROUND}_bench_default.json 2> $O/bench_default.err
cp $(ls -t $R/gpurun_out/bench_detail_*.json | head -1) $O/${ROUND}_bench_default_detail.json
timeout 200 python $R/bench.py --sim hideseek > $O/${ROUND}_bench_hideseek_w8192.json 2> $O/bench_hideseek.err
cp $(ls -t $R/gpurun_out/bench_detail_*.json | head -1) $O/${ROUND}_bench_hideseek_w8192_detail.json
timeout 200 python $R/bench.py --sim escape_room --steps 3000 > $O/${ROUND}_bench_escape_room_w4096.json 2> $O/bench_er.err
cp $(ls -t $R/gpurun_out/bench_detail_*.json | head -1) $O/${ROUND}_bench_escape_room_w4096_detail.json
timeout 200 python $R/bench.py --sim escape_room --worlds 65536 --steps 300 --no-cpu-baseline > $O/${ROUND}_bench_escape_room_w65536.json 2> $O/bench_er64k.err
cp $(ls -t $R/gpurun_out/bench_detail_*.json | head -1) $O/${ROUND}_bench_escape_room_w65536_detail.json
timeout 300 python $R/bench.py --sim escape_room_render > $O/${ROUND}_bench_escape_room_render_w8192.json 2> $O/bench_render.err
cp $(ls -t $R/gpurun_out/bench_detail_*.json | head -1) $O/${ROUND}_bench_escape_room_render_w8192_detail.json
