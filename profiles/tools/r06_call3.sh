#!/bin/bash
# round 6, third GPU call: the whole GPU suite on the pruned code; the physics
# step's phase profile (cycle counters per phase, profile build); SQ issue
# counters of the step kernel at one wavefront per SIMD (two worlds per
# wavefront: the default; the same under a 264-register cap) and at two
# wavefronts per SIMD (one world per wavefront, 256 registers)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_call3.log 2>&1
tail -5 $O/pytest_gpu_call3.log

timeout 300 python profiles/tools/phys_phase_cycles.py 8192 > $O/r06_physics_phases.txt 2> $O/phases.err
sed -i "s/_build_prof'/_build_prof_hh'/" profiles/tools/phys_phase_cycles.py
timeout 300 python profiles/tools/phys_phase_cycles.py 8192 > $O/r06_physics_phases_hullhull.txt 2>> $O/phases.err
sed -i "s/_build_prof_hh'/_build_prof'/" profiles/tools/phys_phase_cycles.py
cat $O/r06_physics_phases.txt | grep -v amdgpu.ids

cd /tmp && export TMPDIR=/tmp
issue() {    # label, env assignments...
  label=$1; shift
  i=0; dbs=""
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
    i=$((i+1))
    rm -rf /tmp/pi_${label}_$i
    env "$@" timeout 400 rocprofv3 --kernel-trace --pmc $set -d /tmp/pi_${label}_$i -o out -- \
        python $R/bench.py --sim escape_room_phys --worlds 8192 --steps 60 --warmup 10 --settle 30 \
        --profile-reps 5 --no-cpu-baseline --no-secondary > /dev/null 2> $O/issue_${label}_$i.err
    dbs="$dbs $(find /tmp/pi_${label}_$i -name '*.db' | head -1)"
  done
  python $R/profiles/tools/make_issue_json.py escape_room_phys 8192 $dbs | \
      python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); d['variant'] = '$label'; print(json.dumps(d))
" >> $O/r06_phys_occupancy_counters.jsonl
}
rm -f $O/r06_phys_occupancy_counters.jsonl
issue lpw32_1wave_421regs DUMMY=1
issue lpw32_1wave_cap264 MADRONA_HIP_BUILD_DIR=_variants/cap132
issue lpw64_2waves_256regs MADRONA_MWHIP_PHYS_LANES=64
grep -c . $O/r06_phys_occupancy_counters.jsonl
python - <<'PY'
import json
for l in open("/root/repo/gpurun_out/r06/r06_phys_occupancy_counters.jsonl"):
    d = json.loads(l)
    if "worldStep(LDS)" in d.get("kernel", ""):
        print(d["variant"], {k: v for k, v in d.items() if k.startswith("SQ_") or k in ("launches",)})
PY
