#!/bin/bash
# Round 5, first GPU call: (1) the driver's own bench command -> is the ONE line
# small and parseable; (2) which instruction-fetch counters this rocprofv3 knows;
# (3) one PMC pass per counter set over the headline workload, summarised per
# launch of the physics step (make_issue_json.py).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err )
wc -c $O/bench_driver_cmd.json
cp $R/gpurun_out/bench_detail.json $O/bench_driver_cmd_detail.json 2>/dev/null

rocprofv3 -L > $O/counters_all.txt 2>&1
grep -o -i -E "\b(SQC?_[A-Z0-9_]*(ICACHE|IFETCH|INST_CACHE|INST_FETCH)[A-Z0-9_]*|SQ_IFETCH[A-Z0-9_]*|SQ_INST_LEVEL_[A-Z0-9_]*|SQ_WAIT_INST_[A-Z0-9_]*|SQ_INSTS_[A-Z0-9_]*|SQ_ACTIVE_INST_[A-Z0-9_]*)\b" $O/counters_all.txt | sort -u > $O/counters_fetch.txt
cat $O/counters_fetch.txt

pass() {   # index, counters...
  i=$1; shift
  rm -rf /tmp/pf_$i
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pf_$i -o out -- \
      python $R/bench.py --steps 60 --warmup 10 --settle 100 --profile-reps 5 --no-cpu-baseline --no-secondary \
      > /dev/null 2> $O/pass_$i.err
  find /tmp/pf_$i -name '*.db' | head -1
}
have() { grep -q -x "$1" $O/counters_fetch.txt && echo "$1"; }
dbs=""
set1="$(have SQC_ICACHE_REQ) $(have SQC_ICACHE_HITS) $(have SQC_ICACHE_MISSES) $(have SQC_ICACHE_MISSES_DUPLICATE) $(have SQ_IFETCH) SQ_WAVES"
echo "set1: $set1"
dbs="$dbs $(pass 1 $set1)"
set2="$(have SQ_WAIT_INST_ANY) $(have SQ_WAIT_INST_LDS) $(have SQ_IFETCH_LEVEL) $(have SQ_INST_LEVEL_VMEM) $(have SQ_INST_LEVEL_LDS) SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
echo "set2: $set2"
dbs="$dbs $(pass 2 $set2)"
set3="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_BUSY_CYCLES"
dbs="$dbs $(pass 3 $set3)"
echo $dbs
python $R/profiles/tools/make_issue_json.py escape_room_phys 8192 $dbs > $O/phys_icache.jsonl
cat $O/phys_icache.jsonl
tail -3 $O/pass_*.err
