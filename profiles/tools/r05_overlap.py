#!/usr/bin/env python3
"""do two half-size executors on two streams fill the physics
step's tail?  One process, one GPU: ONE executor of 8192 Escape-Room+XPBD worlds
against TWO executors of 4096 worlds each (world_base 0 / 4096: the same global
world indices, so the same worlds), each replaying its step graph on its own
stream, the second one started half a step late.  Aggregate env steps/s.

    python profiles/tools/r05_overlap.py [SIM] [WORLDS] >> profiles/r05_overlap.jsonl
"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import bench  # noqa: E402
import torch  # noqa: E402
from madrona_amd.simlib import Simulator, hip_lib_path  # noqa: E402


def run(sim_name, parts, total_worlds, steps=400, settle=400, warmup=50):
    per = total_worlds // parts
    sims = [Simulator(hip_lib_path(sim_name), per, seed=5, gpu_id=0,
                      world_base=i * per, flags=200) for i in range(parts)]
    rings = [bench.fill_actions(sim_name, s, per, 0, 99 + i)
             for i, s in enumerate(sims)]
    for s in sims:
        s.step_async(settle + warmup)
    torch.cuda.synchronize()
    one = None
    if parts > 1:
        # how long one half takes alone (for the stagger and the report)
        t0 = time.perf_counter()
        sims[0].step_async(100)
        torch.cuda.synchronize()
        one = (time.perf_counter() - t0) / 100
        sims[1].step_async(100)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    if parts > 1:
        # stagger: the first executor is half a step ahead
        sims[0].step_async(1)
        time.sleep(one / 2)
        for _ in range(steps - 1):
            for s in sims[1:]:
                s.step_async(1)
            sims[0].step_async(1)
        for s in sims[1:]:
            s.step_async(1)
    else:
        sims[0].step_async(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for s in sims:
        s.sync()
        s.close()
    del rings
    return {"sim": sim_name, "executors": parts, "worlds_each": per,
            "steps": steps, "ms_per_step": dt / steps * 1e3,
            "steps_per_s": total_worlds * steps / dt,
            "one_half_alone_ms": None if one is None else one * 1e3}


if __name__ == "__main__":
    sim_name = sys.argv[1] if len(sys.argv) > 1 else "escape_room_phys"
    worlds = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    torch.cuda.set_device(0)
    for rep in range(2):
        for parts in (1, 2):
            r = run(sim_name, parts, worlds)
            r["repeat"] = rep
            print(json.dumps(r), flush=True)
