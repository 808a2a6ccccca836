#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
for build in _variants/prof_old _build_prof; do
  for lanes in 32 64; do
    name=$(basename $build)_lanes$lanes
    MADRONA_HIP_BUILD_DIR=$build MADRONA_MWHIP_PHYS_LANES=$lanes timeout 300 python profiles/tools/phys_phase_cycles.py 8192 > $O/r06_phases_$name.txt 2>&1
    echo "== $name"; grep -v amdgpu $O/r06_phases_$name.txt | head -13
  done
done
