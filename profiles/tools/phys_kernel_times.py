"""Per-kernel times (dispatch-attached events) of escape_room_phys at N worlds:
    python profiles/tools/phys_kernel_times.py [worlds]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from madrona_amd.simlib import Simulator, hip_lib_path
from collections import defaultdict
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
with Simulator(hip_lib_path('escape_room_phys'), W, seed=5, flags=200) as hip:
    rng = np.random.default_rng(0)
    a = np.stack([rng.integers(0, 4, (W, 2)), rng.integers(0, 8, (W, 2)),
                  rng.integers(-2, 3, (W, 2)), rng.integers(0, 2, (W, 2))], -1).astype(np.int32)
    hip.write_tensor('action', a)
    hip.step(100)
    t = time.perf_counter(); hip.step(300); dt = time.perf_counter() - t
    print(f'{os.environ.get("MADRONA_HIP_BUILD_DIR","_build")}: {dt/300*1e6:.0f} us/step, {W*300/dt/1e6:.2f} M steps/s')
    agg = defaultdict(float)
    for k in hip.profile(10):
        agg[k['name']] += k['avg_us']
    for n, v in sorted(agg.items(), key=lambda x: -x[1])[:8]:
        print(f'   {v:9.1f} {n}')
