import sys; sys.path.insert(0,'.')
from madrona_amd.simlib import Simulator, hip_lib_path, runtime_lib
import bench, torch, ctypes as C
worlds=8192
rt = runtime_lib()
rt.mwhip_num_table_growths.restype = C.c_uint32
rt.mwhip_num_table_growths.argtypes = [C.c_void_p]
with Simulator(hip_lib_path("escape_room_render"), worlds, seed=5, flags=200 | (64 << 16)) as sim:
    bench.fill_actions("escape_room_phys", sim, worlds, 0, 77)
    rg = sim.render_graph()
    print("growths after create", rt.mwhip_num_table_growths(sim.hip_exec()))
    for i in range(6):
        for _ in range(20):
            sim.step_async(1); sim.step_async(1, graph=rg)
        sim.sync()
        print("after", (i+1)*20, "steps: growths", rt.mwhip_num_table_growths(sim.hip_exec()))
    st = sim.profile(10)
    print([ (k["name"], round(k["avg_us"],1)) for k in st if "grab" in k["name"]])
