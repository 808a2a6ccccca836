"""Per-kernel times of escape_room (config 2) at N worlds:
    python profiles/tools/er_kernel_times.py [worlds]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from madrona_amd.simlib import Simulator, hip_lib_path
from collections import defaultdict
W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
with Simulator(hip_lib_path('escape_room'), W, seed=5, flags=200) as hip:
    rng = np.random.default_rng(0)
    a = np.stack([rng.integers(0, 4, (W, 2)), rng.integers(0, 8, (W, 2)),
                  rng.integers(-2, 3, (W, 2)), np.zeros((W, 2), int)], -1).astype(np.int32)
    hip.write_tensor('action', a)
    hip.step(200)
    t = time.perf_counter(); hip.step_async(1000); hip.sync(); dt = time.perf_counter() - t
    print(f'{os.environ.get("MADRONA_HIP_BUILD_DIR","_build")}: {dt/1000*1e6:.1f} us/step, {W*1000/dt/1e6:.2f} M steps/s')
    agg = defaultdict(float)
    for k in hip.profile(20):
        agg[k['name']] += k['avg_us']
    for n, v in sorted(agg.items(), key=lambda x: -x[1])[:4]:
        print(f'   {v:9.1f} {n}')
