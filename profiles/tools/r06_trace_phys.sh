#!/bin/bash
# Device trace of the headline step (MADRONA_TRACING build): when the physics
# kernel's workgroups start and finish.  OUT=gpurun_out/trace
set -u
OUT=${OUT:-gpurun_out/trace}
SIM=${SIM:-escape_room_phys}
mkdir -p $OUT
export MADRONA_HIP_BUILD_DIR=${BUILD:-_build_tracing} MADRONA_MWHIP_TRACE_DIR=$OUT MADRONA_MWGPU_TRACE_NAME=phys
timeout 300 python profiles/tools/trace_sim.py $SIM 8192 2 ${STEPS:-24} > $OUT/run.log 2>&1
LOG=$(ls $OUT/*_madrona_device_tracing.bin | head -1)
python madrona_amd/scripts/parse_device_tracing.py $LOG > $OUT/nodes.txt 2>&1
N=$(grep "worldStep(LDS)" $OUT/nodes.txt | head -1 | awk '{print $1}')
for st in -1 -4 -8; do
  python madrona_amd/scripts/parse_device_tracing.py $LOG --step $st --node ${N:-9} | tail -7 > $OUT/phys_step$st.txt 2>&1
done
python profiles/tools/trace_wgs.py $LOG ${N:-9} > $OUT/phys_wgs.txt 2>&1
rm -f $OUT/*.bin
cat $OUT/phys_step-1.txt $OUT/phys_step-4.txt $OUT/phys_wgs.txt
