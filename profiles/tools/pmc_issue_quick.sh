#!/bin/bash
# Instruction-issue counters of the headline workload's kernels (two PMC passes,
# kernel trace only), summarised into gpurun_out/issue/.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/issue
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rm -rf /tmp/pi_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pi_$i -o out -- \
      python $R/bench.py --steps 60 --warmup 10 --settle 100 --no-cpu-baseline --no-secondary \
      > /dev/null 2> $O/pass_$i.err
  eval "db_$i=$(find /tmp/pi_$i -name '*.db' | head -1)"
done
python $R/profiles/tools/make_issue_json.py escape_room_phys 8192 $db_1 $db_2 > $O/issue_escape_room_phys_8192.jsonl
cat $O/issue_escape_room_phys_8192.jsonl
