"""debug: hideseek world 601 alone"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from madrona_amd.simlib import Simulator, hip_lib_path, ref_lib_path
acts = np.load('tests/golden/dbg_actions.npy')
np.set_printoptions(precision=9, linewidth=200, suppress=True)
with Simulator(ref_lib_path('hideseek'), 1, seed=5, num_workers=1, flags=200, world_base=601) as ref, \
        Simulator(hip_lib_path('hideseek'), 1, seed=5, flags=200, world_base=601) as hip:
    for s in range(1, 26):
        a = acts[s - 1][None]
        ref.write_tensor('action', a); hip.write_tensor('action', a)
        if s == 24:
            r = ref.dump_all()
            for col in ('Agent.Position', 'Agent.Rotation', 'MovableObject.Position', 'MovableObject.Rotation', 'MovableObject.Velocity', 'MovableObject.ResponseType', 'StaticObject.Position', 'StaticObject.Scale'):
                v = r[col][0]
                print('pre24', col); print(v.view(np.float32) if 'Resp' not in col else v.view(np.int32).T)
        ref.step(1); hip.step(1)
        r = ref.dump_all(); h = hip.dump_all()
        for col in r:
            rr, rc = r[col]; hr, hc = h[col]
            if not np.array_equal(rr, hr):
                diff = np.nonzero((rr != hr).any(axis=1))[0]
                print('step', s, col, 'rows', list(diff[:8]))
                if 'Lidar' not in col and 'Obs' not in col:
                    for d in diff[:4]:
                        print('   ref', rr[d].view(np.float32), '\n   hip', hr[d].view(np.float32))
