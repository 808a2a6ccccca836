#!/bin/bash
# Regenerates the round's measurements on the GPU box (run through gpurun from
# the repo root): bench lines, rocprofv3 kernel stats of the same commands
# (trimmed to the timed window by the benchWindowMarker dispatches), and the
# FETCH_SIZE / WRITE_SIZE counter passes (one counter per pass, kernel trace
# only).  Outputs land in gpurun_out/refresh; copy what should be judged into
# profiles/ (ROUND = file prefix).
set -x
ROUND=${ROUND:-r02}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh
mkdir -p $O
cd /tmp && export TMPDIR=/tmp

# the driver's command line, then the other BASELINE shapes
python $R/bench.py > $O/${ROUND}_bench_default.json 2> $O/bench_default.err
python $R/bench.py --sim escape_room --steps 3000 > $O/${ROUND}_bench_escape_room_w4096.json 2> $O/bench_er.err
python $R/bench.py --sim hideseek > $O/${ROUND}_bench_hideseek_w8192.json 2> $O/bench_hideseek.err
python $R/bench.py --sim escape_room --worlds 65536 --steps 300 --no-cpu-baseline > $O/${ROUND}_bench_escape_room_w65536.json 2> $O/bench_er64k.err

prof() {   # name, bench args...
  name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o out -- \
      python $R/bench.py "$@" --warmup 100 --no-cpu-baseline --no-secondary \
      > $O/${ROUND}_${name}_under_rocprof.json 2> $O/${name}_under_rocprof.err
  db=$(find /tmp/prof_$name -name '*.db' | head -1)
  python $R/profiles/summarize_rocprof.py $db $O/${ROUND}_${name}_kernel_stats
}
prof bench_escape_room_phys_w8192 --steps 300
prof bench_escape_room_w4096 --sim escape_room --steps 1000
prof bench_hideseek_w8192 --sim hideseek --steps 300

pmc() {    # sim, worlds, bench args...
  sim=$1; worlds=$2; shift; shift
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${sim}_${worlds}_$ctr
    timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_${sim}_${worlds}_$ctr -o out -- \
        python $R/bench.py --sim $sim --worlds $worlds "$@" --steps 100 --warmup 20 --settle 30 \
        --no-cpu-baseline --no-secondary > /dev/null 2> $O/pmc_${sim}_${worlds}_$ctr.err
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
python - <<PYEOF
import glob, json
entries = []
for f in sorted(glob.glob("$O/traffic_*.json")):
    entries += json.load(open(f))
json.dump({"_comment": "HBM traffic per STEP of every kernel from rocprofv3 --pmc FETCH_SIZE / "
           "WRITE_SIZE (separate passes, profiles/tools/refresh_profiles.sh + make_traffic_json.py), "
           "averaged over a 120-replay bench run + its per-kernel profiling replays. bytes = 2 * "
           "FETCH_SIZE KiB (gfx950 reports half of the fetched bytes, MI355X_MICROARCH.md) + "
           "WRITE_SIZE KiB. 'step:all-kernels' = sum over every kernel of a replay. bench.py "
           "copies matching entries into roofline.*.traffic.",
           "entries": entries}, open("$O/${ROUND}_hbm_traffic.json", "w"), indent=1)
PYEOF
ls -la $O
