#!/usr/bin/env python3
"""Runs the render graph of sims/escape_room_render a few times (for rocprofv3
--pmc passes over the ray caster: python render_kernel_times.py WORLDS REPS
[RES] [SHADOWS])."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from madrona_amd.simlib import Simulator, hip_lib_path  # noqa: E402

worlds = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
res = int(sys.argv[3]) if len(sys.argv) > 3 else 64
shadows = int(sys.argv[4]) if len(sys.argv) > 4 else 0
with Simulator(hip_lib_path("escape_room_render"), worlds, seed=5,
               flags=200 | (res << 16) | (shadows << 25)) as sim:
    sim.step(20)
    for _ in range(reps):
        sim.render()
    stats = sim.profile(5, graph=sim.render_graph())
    for k in stats:
        print(k["name"], round(k["avg_us"], 1), "us")
