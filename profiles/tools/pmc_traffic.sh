set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmc/er_$C -o out -- python $R/bench.py --steps 100 --warmup 20 --no-physics-line --no-cpu-baseline > $R/gpurun_out/pmc/er_$C.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmc/er64k_$C -o out -- python $R/bench.py --steps 50 --warmup 10 --worlds 65536 --no-physics-line --no-cpu-baseline > $R/gpurun_out/pmc/er64k_$C.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmc/phys_$C -o out -- python $R/bench.py --sim escape_room_phys --steps 100 --warmup 20 --no-cpu-baseline > $R/gpurun_out/pmc/phys_$C.log 2>&1
done
cd $R
for d in gpurun_out/pmc/*/; do
  db=$(find $d -name '*.db' | head -1)
  echo "== $d"; python profiles/summarize_pmc.py $db > ${d%/}.summary.txt 2>&1; true
done
find gpurun_out/pmc -name "*.db" -delete; ls -la gpurun_out/pmc
