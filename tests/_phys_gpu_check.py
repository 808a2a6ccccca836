"""Scratch: lock-step escape_room_phys on ref CPU vs HIP, report first mismatch."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.parity_utils import compare_columns
from madrona_amd.simlib import Simulator, hip_lib_path, ref_lib_path

W = int(sys.argv[1]) if len(sys.argv) > 1 else 4
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 60
FLAGS = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rng = np.random.default_rng(1)
with Simulator(ref_lib_path('escape_room_phys'), W, seed=5, num_workers=1, flags=FLAGS) as ref, \
        Simulator(hip_lib_path('escape_room_phys'), W, seed=5, flags=FLAGS) as hip:
    probs = compare_columns(ref.dump_all(), hip.dump_all())
    print('init', probs[:6])
    for s in range(1, STEPS + 1):
        a = ref.read_tensor('action')
        a[..., 0] = rng.integers(0, 4, a.shape[:-1])
        a[..., 1] = rng.integers(0, 8, a.shape[:-1])
        a[..., 2] = rng.integers(-2, 3, a.shape[:-1])
        a[..., 3] = rng.integers(0, 2, a.shape[:-1])
        ref.write_tensor('action', a)
        hip.write_tensor('action', a)
        ref.step(1)
        hip.step(1)
        probs = compare_columns(ref.dump_all(), hip.dump_all())
        # observations of step 0 are uninitialised in the reference
        if probs:
            print('step', s, 'MISMATCH')
            for p in probs[:12]:
                print('  ', p)
            rd, hd = ref.dump_all(), hip.dump_all()
            for name in ('Agent.Position', 'PhysicsEntity.Position'):
                r = rd[name][0].view(np.float32); h = hd[name][0].view(np.float32)
                if r.shape == h.shape:
                    d = np.abs(r - h).max(axis=1)
                    bad = np.nonzero(d > 0)[0]
                    print(name, 'rows differing', bad[:10], 'max abs', d.max())
                    for b in bad[:4]:
                        print('   ref', r[b], 'hip', h[b])
            break
    else:
        print('all', STEPS, 'steps match')
    err = hip.error_flags() if hasattr(hip, 'error_flags') else None
    print('error flags', err)
