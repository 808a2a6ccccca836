#!/bin/bash
# Round-6 measurements on the GPU box (through gpurun, from the repo root):
#   bench lines (driver command, hideseek, escape_room 4096 / 65536, config 5),
#   rocprofv3 --kernel-trace --stats of the headline and of the config-5 render
#   pass, both trimmed to the timed window by the benchWindowMarker dispatches,
#   FETCH_SIZE / WRITE_SIZE passes (one counter per pass, kernel trace only),
#   SQ instruction-issue counters (two passes) of the headline and of config 5.
# Outputs: gpurun_out/refresh/r03_*; copy what should be judged into profiles/.
set -x
ROUND=${ROUND:-r06}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh
mkdir -p $O
cd /tmp && export TMPDIR=/tmp

pmc() {    # sim, worlds, bench args...
  sim=$1; worlds=$2; shift; shift
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${sim}_${worlds}_$ctr
    timeout 400 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_${sim}_${worlds}_$ctr -o out -- \
        python $R/bench.py --sim $sim --worlds $worlds "$@" --steps 60 --warmup 10 --settle 30 \
        --profile-reps 5 --no-cpu-baseline --no-secondary > /dev/null 2> $O/pmc_${sim}_${worlds}_$ctr.err
    db=$(find /tmp/pmc_${sim}_${worlds}_$ctr -name '*.db' | head -1)
    python $R/profiles/summarize_pmc.py $db > $O/${ROUND}_pmc_${sim}_w${worlds}_$ctr.txt
    eval "db_$ctr=$db"
  done
  python $R/profiles/tools/make_traffic_json.py $sim $worlds $db_FETCH_SIZE $db_WRITE_SIZE \
      > $O/traffic_${sim}_${worlds}.json
}
pmc escape_room_phys 8192
pmc escape_room 4096
pmc escape_room 65536
pmc hideseek 8192
pmc escape_room_render 8192
python - <<PYEOF
import glob, json
entries = []
for f in sorted(glob.glob("$O/traffic_*.json")):
    entries += json.load(open(f))
json.dump({"_comment": "HBM traffic per STEP of every kernel from rocprofv3 --pmc FETCH_SIZE / "
           "WRITE_SIZE (separate passes, profiles/tools/refresh_r06.sh + make_traffic_json.py), "
           "averaged over a 70-replay bench run + its per-kernel profiling replays, all from the "
           "final code of round 6. bytes = 2 * FETCH_SIZE KiB (gfx950 reports half of the fetched "
           "bytes, MI355X_MICROARCH.md) + WRITE_SIZE KiB. 'step:all-kernels' = sum over every "
           "kernel of a replay. bench.py copies matching entries into roofline.*.traffic.",
           "entries": entries}, open("$O/${ROUND}_hbm_traffic.json", "w"), indent=1)
PYEOF

issue() {    # sim, worlds, bench args...
  sim=$1; worlds=$2; shift; shift
  i=0; dbs=""
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
    i=$((i+1))
    rm -rf /tmp/pi_${sim}_$i
    timeout 400 rocprofv3 --kernel-trace --pmc $set -d /tmp/pi_${sim}_$i -o out -- \
        python $R/bench.py --sim $sim --worlds $worlds "$@" --steps 60 --warmup 10 --settle 30 \
        --profile-reps 5 --no-cpu-baseline --no-secondary > /dev/null 2> $O/issue_${sim}_$i.err
    dbs="$dbs $(find /tmp/pi_${sim}_$i -name '*.db' | head -1)"
  done
  python $R/profiles/tools/make_issue_json.py $sim $worlds $dbs > $O/issue_${sim}_${worlds}.jsonl
}
issue escape_room_phys 8192
issue hideseek 8192
issue escape_room_render 8192
python - <<PYEOF
import glob, json
entries = []
for f in sorted(glob.glob("$O/issue_*.jsonl")):
    entries += [json.loads(l) for l in open(f) if l.startswith("{")]
json.dump({"_comment": "SQ instruction / cycle counters per LAUNCH (rocprofv3 --pmc, kernel trace "
           "only, two passes; profiles/tools/refresh_r06.sh + make_issue_json.py): SQ_INSTS_* = "
           "wave-instructions issued, SQ_WAVE_CYCLES / SQ_ACTIVE_INST_ANY / SQ_WAIT_ANY / "
           "SQ_WAIT_INST_ANY in quad-cycles summed over waves. bench.py turns them into the "
           "valu-issue rooflines with the live kernel durations.",
           "entries": entries}, open("$O/${ROUND}_issue_counters.json", "w"), indent=1)
PYEOF
ls -la $O

# bench.py copies the recorded traffic / issue counters into its rooflines: the
# files of THIS refresh, so they go where it looks before the bench lines run
cp $O/${ROUND}_hbm_traffic.json $O/${ROUND}_issue_counters.json $R/profiles/ 2>/dev/null

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

prof() {   # name, command...
  name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o out -- "$@" \
      > $O/${ROUND}_${name}_under_rocprof.json 2> $O/${name}_under_rocprof.err
  db=$(find /tmp/prof_$name -name '*.db' | head -1)
  python $R/profiles/summarize_rocprof.py $db $O/${ROUND}_${name}_kernel_stats
}
prof bench_escape_room_phys_w8192 python $R/bench.py --steps 300 --warmup 100 --no-cpu-baseline --no-secondary
prof render_config5_w8192 python $R/bench.py --sim escape_room_render --steps 100 --warmup 20 --no-cpu-baseline
prof bench_hideseek_w8192 python $R/bench.py --sim hideseek --steps 300 --warmup 100 --no-cpu-baseline --no-secondary

