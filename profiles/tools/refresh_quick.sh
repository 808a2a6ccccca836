#!/bin/bash
# The short version of refresh_profiles.sh (a few GPU-minutes): the driver's
# bench line, rocprofv3 kernel stats of the headline and of the config-5 render
# pass, the atomic microbenchmark, and the two PMC passes over the headline.
# Outputs in gpurun_out/refresh; copy what should be judged into profiles/.
set -x
ROUND=${ROUND:-r02}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh
mkdir -p $O
cd /tmp && export TMPDIR=/tmp

timeout 300 python $R/bench.py > $O/${ROUND}_bench_default.json 2> $O/bench_default.err
timeout 200 python $R/bench.py --sim hideseek > $O/${ROUND}_bench_hideseek_w8192.json 2> $O/bench_hideseek.err

/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_mb $R/profiles/tools/atomic_microbench.hip \
  && timeout 120 /tmp/atomic_mb > $O/${ROUND}_atomic_microbench.txt 2>&1

prof() {   # name, command...
  name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o out -- "$@" \
      > $O/${ROUND}_${name}_under_rocprof.json 2> $O/${name}_under_rocprof.err
  db=$(find /tmp/prof_$name -name '*.db' | head -1)
  python $R/profiles/summarize_rocprof.py $db $O/${ROUND}_${name}_kernel_stats
}
prof bench_escape_room_phys_w8192 python $R/bench.py --steps 300 --warmup 100 --no-cpu-baseline --no-secondary
prof render_config5_w8192 python -c "
import sys, json; sys.path.insert(0, '$R')
import bench
print(json.dumps(bench.run_render(8192, 0, 5, 200, 100, 20, 5, 100)))"

for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  timeout 400 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_$ctr -o out -- \
      python $R/bench.py --steps 100 --warmup 20 --settle 30 --no-cpu-baseline --no-secondary \
      > /dev/null 2> $O/pmc_$ctr.err
  db=$(find /tmp/pmc_$ctr -name '*.db' | head -1)
  python $R/profiles/summarize_pmc.py $db > $O/${ROUND}_pmc_escape_room_phys_w8192_$ctr.txt
  eval "db_$ctr=$db"
done
python $R/profiles/tools/make_traffic_json.py escape_room_phys 8192 $db_FETCH_SIZE $db_WRITE_SIZE \
    > $O/traffic_escape_room_phys_8192_final.json
ls -la $O
