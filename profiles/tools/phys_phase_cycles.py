"""Per-phase cycle counters of the fused physics step (needs a profile build:

    make -C madrona_amd OUT=_build_prof EXTRA=-DMADRONA_PHYS_PROFILE=1 runtime \\
         _build_prof/libescape_room_phys_hip.so

the kernel then adds s_memtime deltas per phase to ecs_state::moduleData[1])."""
import sys, os, ctypes as C
os.environ.setdefault('MADRONA_HIP_BUILD_DIR', '_build_prof')
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from madrona_amd.simlib import Simulator, hip_lib_path, runtime_lib
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
with Simulator(hip_lib_path('escape_room_phys'), W, seed=5, flags=200) as hip:
    rt = runtime_lib()
    rt.mwhip_alloc_device.restype = C.c_void_p
    rt.mwhip_alloc_device.argtypes = [C.c_void_p, C.c_uint64, C.c_int]
    rt.mwhip_set_module_data.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    rt.mwhip_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    import torch
    rng = np.random.default_rng(0)
    # a new action set every step (the bench's input ring)
    ring = np.stack([np.stack([rng.integers(0, 4, (W, 2)), rng.integers(0, 8, (W, 2)),
                               rng.integers(-2, 3, (W, 2)), rng.integers(0, 2, (W, 2))], -1)
                     for _ in range(61)]).astype(np.int32)
    dev = torch.from_numpy(ring).cuda()
    hip.set_input_ring('action', dev.data_ptr(), 61)
    hip.step(300)
    buf = rt.mwhip_alloc_device(hip.hip_exec(), 256, 1)
    rt.mwhip_set_module_data(hip.hip_exec(), 1, buf)
    N = 50
    hip.step(N)
    out = np.zeros(32, np.uint64)
    rt.mwhip_memcpy_d2h(out.ctypes.data, buf, 256)
    if os.environ.get('PHYS_PROFILE_HH_IN_LDS'):
        # (-DMADRONA_PHYS_PROFILE_LDS_HH: the hull-hull stages sit in slots 0..8)
        hh, events, out = np.concatenate([out[:9], np.zeros(7, np.uint64)]), out[12:16], np.zeros(12, np.uint64)
        out[0] = 1
    else:
        hh, events, out = out[16:], out[12:16], out[:12]
    # (slot 2 also collects the velocity solve of the substep before it, slot 6
    # only that of the last substep: see the PHYS_PROF marks in world_step.inl)
    names = ['np.setup', 'candidates', 'integrate', 'np.solo', 'solvePos+jnt+setVel', 'np.hull+compact', '(end of substeps)', 'joints / store (+ refit)', 'load bodies (rows -> block)', 'stage prims', 'world lookup (singletons, row ranges)', 'solveVel']
    tot = out.sum()
    for n, v in zip(names, out):
        print(f'{n:42s} {v / N / W:10.0f} cycles per wavefront and step  {100 * v / tot:5.1f}%')
    print('total cycles per wavefront and step (shader clock)', tot / N / W)
    # event counts of the same steps (lane 0 of every world)
    hull, hull_hit, solo, cands = (float(v) / N / W for v in events)
    print(f'per world and step (4 substeps): {cands:.1f} candidate tests, {solo:.1f} per-lane '
          f'pairs, {hull:.1f} hull-hull pairs of which {hull_hit:.1f} touch')
    # stages of the cooperative hull-hull tests (HullHullProf, world_step.inl)
    hh = hh.astype(np.float64) / N / W
    tests = hh[5] + hh[6] + hh[7] + hh[8]
    print(f'hull-hull tests per world and step: {tests:.2f}; separated by a face of A '
          f'{hh[5]:.2f}, by a face of B {hh[6]:.2f}, by an edge pair {hh[7]:.2f}, '
          f'overlapping {hh[8]:.2f}')
    for n, v in zip(['hulls -> LDS', 'face query A', 'face query B', 'edge query',
                     'manifold'], hh[:5]):
        print(f'  hull-hull {n:14s} {v:10.0f} cycles per wavefront and step')
