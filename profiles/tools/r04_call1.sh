#!/bin/bash
# round 4, GPU call 1: hull-hull stage profile, register-cap variant, the ring
# test, the 65536-world gather
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 300 python profiles/tools/phys_phase_cycles.py 8192 > gpurun_out/r04_physics_phases_hh.txt 2>&1; echo "phases rc=$?")
(timeout 400 python profiles/tools/build_variants.py escape_room_phys 8192 worldStep _build > gpurun_out/r04_phys_variants_call1.jsonl 2>&1; echo "variants rc=$?")
(timeout 300 python -m pytest tests/test_escape_room_render_gpu.py::test_input_ring_ignores_render_replays tests/test_parity_gpu.py -k "input_ring" -x -q > gpurun_out/r04_ring_tests.txt 2>&1; echo "ring tests rc=$?"; tail -3 gpurun_out/r04_ring_tests.txt)
(timeout 300 python bench.py --sim escape_room --worlds 65536 --no-cpu-baseline --no-secondary > gpurun_out/r04_bench_escape_room_w65536_call1.json 2> gpurun_out/r04_bench_w65536_call1.err; echo "bench65536 rc=$?")
cat gpurun_out/r04_physics_phases_hh.txt | tail -25
cat gpurun_out/r04_phys_variants_call1.jsonl
python - <<'PY'
import json
try:
    r = json.loads([l for l in open('gpurun_out/r04_bench_escape_room_w65536_call1.json') if l.startswith('{')][-1])
    print('w65536', r['value'], r['ms_per_step'])
    for k in r.get('kernels', []):
        if 'sort' in k['name'].lower():
            print(k['name'], k.get('role'), k['avg_us'])
    print(json.dumps(r['roofline'].get('nodes', {}).get('sort_node', {}))[:1500])
except Exception as e:
    print('bench parse failed', e)
PY
