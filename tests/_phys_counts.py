import sys, os, ctypes as C
os.environ['MADRONA_MWHIP_PHYS_KEEP_CONTACTS'] = '1'
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madrona_amd.simlib import Simulator, hip_lib_path, runtime_lib
W = 1
ARCH, COMP = int(sys.argv[1]), int(sys.argv[2])
with Simulator(hip_lib_path('escape_room_phys'), W, seed=5) as hip:
    hip.step(30)
    rt = runtime_lib()
    rt.mwhip_dump_column.restype = C.c_int64
    rt.mwhip_dump_column.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]
    buf = np.zeros(100 * 1024, np.uint8); counts = np.zeros(W, np.int32)
    n = rt.mwhip_dump_column(hip.hip_exec(), ARCH, COMP, buf.ctypes.data, buf.nbytes, counts.ctypes.data)
    print('contacts', n, counts)
    rows = buf[:n * 100].reshape(n, 100)
    locs = rows[:, :16].copy().view(np.int32)
    npts = rows[:, 80:84].copy().view(np.int32)
    nrm = rows[:, 84:96].copy().view(np.float32)
    pts = rows[:, 16:80].copy().view(np.float32)
    for i in range(n):
        print(locs[i], npts[i], nrm[i].round(3), pts[i][:4].round(3))
