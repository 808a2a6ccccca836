import sys; sys.path.insert(0,'.')
from madrona_amd.simlib import Simulator, hip_lib_path
import bench, torch
worlds=8192
with Simulator(hip_lib_path("escape_room_render"), worlds, seed=5, flags=200 | (64 << 16)) as sim:
    bench.fill_actions("escape_room_phys", sim, worlds, 0, 77)
    sim.step_async(300); sim.sync()
    st = sim.profile(20)
    tot = sum(k["avg_us"] for k in st)
    print("total", round(tot,1), "kernels", len(st))
    for k in sorted(st, key=lambda k: -k["avg_us"])[:40]:
        print(f'{k["avg_us"]:9.1f}  {k["rows"]:10.0f}  {k["name"]}')
