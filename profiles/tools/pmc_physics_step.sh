cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -io "SQC\?_[A-Z_]*\(IFETCH\|ICACHE\|INST_LEVEL\|WAIT_INST\)[A-Z_0-9]*" | sort -u | tr '\n' ' '
echo
R=$GRAFT_REPO_ROOT
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  rm -rf /tmp/p; timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/p -o out -- python $R/profiles/tools/phys_kernel_times.py 2048 > /tmp/p.log 2>&1
  db=$(find /tmp/p -name '*.db' | head -1)
  python $R/profiles/summarize_pmc.py $db "physicsStep" 
done
