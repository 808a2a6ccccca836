import numpy as np, sys, os
sys.path.insert(0, ".")
from madrona_amd.simlib import Simulator, hip_lib_path
import torch
W = 8192
with Simulator(hip_lib_path("escape_room_phys"), W, seed=1, flags=200) as s:
    rng = np.random.default_rng(0)
    ring = np.stack([np.stack([rng.integers(0, 4, (W, 2)), rng.integers(0, 8, (W, 2)), rng.integers(-2, 3, (W, 2)), rng.integers(0, 2, (W, 2))], -1) for _ in range(61)]).astype(np.int32)
    dev = torch.from_numpy(ring).cuda()
    s.set_input_ring("action", dev.data_ptr(), 61)
    s.step(int(sys.argv[1]))
