#!/bin/bash
# Regenerates the round's measurements on the GPU box (run through gpurun from
# the repo root): bench lines, rocprofv3 kernel stats of the same commands, and
# the FETCH_SIZE / WRITE_SIZE counter passes.  Outputs land in gpurun_out/refresh;
# copy what should be judged into profiles/.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh
mkdir -p $O
cd /tmp && export TMPDIR=/tmp

python $R/bench.py > $O/bench_escape_room_w4096.json 2> $O/bench_escape_room_w4096.err
python $R/bench.py --sim escape_room_phys > $O/bench_escape_room_phys_w8192.json 2> $O/bench_phys.err
python $R/bench.py --sim hideseek > $O/bench_hideseek_w8192.json 2> $O/bench_hideseek.err

prof() {   # name, bench args...
  name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o out -- \
      python $R/bench.py "$@" --steps 300 --warmup 100 --no-cpu-baseline --no-physics-line \
      > $O/${name}_under_rocprof.json 2> $O/${name}_under_rocprof.err
  db=$(find /tmp/prof_$name -name '*.db' | head -1)
  python $R/profiles/summarize_rocprof.py $db $O/${name}_kernel_stats
}
prof bench_escape_room_w4096
prof bench_escape_room_phys_w8192 --sim escape_room_phys
prof bench_hideseek_w8192 --sim hideseek

pmc() {    # name, counter, bench args...
  name=$1; ctr=$2; shift; shift
  rm -rf /tmp/pmc_$name
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_$name -o out -- \
      python $R/bench.py "$@" --steps 100 --warmup 20 --no-cpu-baseline --no-physics-line \
      > /dev/null 2> $O/pmc_${name}_$ctr.err
  db=$(find /tmp/pmc_$name -name '*.db' | head -1)
  python $R/profiles/summarize_pmc.py $db > $O/pmc_${name}_$ctr.txt
}
for C in FETCH_SIZE WRITE_SIZE; do
  pmc escape_room_w4096 $C
  pmc escape_room_w65536 $C --worlds 65536
  pmc escape_room_phys_w8192 $C --sim escape_room_phys
  pmc hideseek_w8192 $C --sim hideseek
done
ls -la $O
