import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from madrona_amd.simlib import Simulator, hip_lib_path
sim = sys.argv[1]; W = int(sys.argv[2]); flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
with Simulator(hip_lib_path(sim), W, flags=flags) as s:
    s.step(20)
    t0 = time.time(); s.step(200); dt = time.time() - t0
    print(f"[{sim}] W={W}: {dt/200*1e6:.1f} us/step, {W*200/dt/1e6:.2f} M steps/s")
    st = s.profile(20)
    tot = sum(k['avg_us'] for k in st)
    print(f"profiled total {tot:.1f} us over {len(st)} kernels")
    for k in st:
        gbs = k['algo_bytes'] / (k['avg_us'] * 1e-6) / 1e9 if k['avg_us'] > 0 else 0
        print(f"  {k['avg_us']:8.2f} us  rows={k['rows']:10.0f}  algoMB={k['algo_bytes']/1e6:8.3f}  {gbs:8.1f} GB/s  {k['name']}")
