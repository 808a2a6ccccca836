#!/usr/bin/env python3
"""Runs a simulator of a MADRONA_TRACING build (MADRONA_HIP_BUILD_DIR=_build_tracing)
for N steps with a fresh action set every step, so that the executor writes its
device event log on exit:  trace_sim.py SIM WORLDS AGENTS STEPS"""
import sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import torch
from madrona_amd.simlib import Simulator, hip_lib_path

sim, W, A, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
with Simulator(hip_lib_path(sim), W, seed=1, flags=200) as s:
    rng = np.random.default_rng(0)
    ring = np.stack([np.stack([rng.integers(0, 4, (W, A)), rng.integers(0, 8, (W, A)),
                               rng.integers(-2, 3, (W, A)), rng.integers(0, 2, (W, A))], -1)
                     for _ in range(61)]).astype(np.int32)
    dev = torch.from_numpy(ring).cuda()
    s.set_input_ring("action", dev.data_ptr(), 61)
    s.step(steps)
