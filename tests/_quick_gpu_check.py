import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from parity_utils import run_pair, Simulator, hip_lib_path
sim = sys.argv[1] if len(sys.argv) > 1 else 'cartpole'
W = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
t0 = time.time()
flags = int(sys.argv[4]) if len(sys.argv) > 4 else 0
rng = np.random.default_rng(0)
def actions(ref, hip, s):
    if sim != 'escape_room': return
    a = np.stack([rng.integers(0,4,(W,2)), rng.integers(0,8,(W,2)), rng.integers(-2,3,(W,2)), np.zeros((W,2),int)], -1).astype(np.int32)
    ref.write_tensor('action', a); hip.write_tensor('action', a)
probs, step = run_pair(sim, W, steps, flags=flags, actions=actions, check_init=(sim != 'escape_room'))
print(f"[{sim}] W={W} steps={steps} parity problems at step {step}: {probs[:5]} ({time.time()-t0:.1f}s)")
with Simulator(hip_lib_path(sim), 4096, flags=flags) as s:
    s.step(10)
    t0 = time.time(); s.step(200); dt = time.time() - t0
    print(f"[{sim}] hip 4096 worlds: {200/dt:.0f} graph replays/s, {4096*200/dt/1e6:.2f} M steps/s")
