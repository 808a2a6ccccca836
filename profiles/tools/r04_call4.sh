#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
(timeout 600 python profiles/tools/run_variants.py profiles/variants/r04_phys_centroid.json 2 > gpurun_out/r04_phys_centroid.jsonl 2>&1; echo "variants rc=$?")
cut -c1-330 gpurun_out/r04_phys_centroid.jsonl
(timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_physics_functions_gpu.py -x -q -k "portable or lanes_per_world or escape_room_physics_lockstep or hideseek_lockstep or ball_pit_lockstep or physics_functions or pair" > gpurun_out/r04_tests_call4.txt 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r04_tests_call4.txt)
(timeout 600 python bench.py > gpurun_out/r04_bench_default_call4.json 2> gpurun_out/r04_bench_default_call4.err; echo "bench rc=$?")
python - <<'PY'
import json
r = json.loads([l for l in open('gpurun_out/r04_bench_default_call4.json') if l.startswith('{')][-1])
print('headline', r['value'], r['ms_per_step'])
n = r['roofline']['nodes']
for k in ('parallel_for', 'parallel_for_signature_rule', 'sort_and_parallel_for', 'sort_node'):
    if k in n:
        e = dict(n[k]); ks = e.pop('kernels', None); e.pop('note', None); e.pop('chains', None)
        print(k, json.dumps(e))
        for x in ks or []:
            print('    ', x)
print('portable', json.dumps(r.get('portable_sim'))[:1500])
e2 = r['ecs_config2']; print('config2', e2['value'], e2['ms_per_step'])
n2 = e2['roofline']['nodes']
for k in ('parallel_for', 'parallel_for_signature_rule', 'sort_and_parallel_for', 'sort_node'):
    if k in n2:
        e = dict(n2[k]); ks = e.pop('kernels', None); e.pop('note', None); e.pop('chains', None)
        print(k, json.dumps(e))
        for x in ks or []:
            print('    ', x)
PY
