#!/usr/bin/env python3
"""HBM traffic per STEP of every kernel, from two rocprofv3 PMC passes of the
same bench command (FETCH_SIZE, WRITE_SIZE; one counter per pass, kernel trace
only -- MI355X_MICROARCH.md, HBM section) -> entries for
profiles/rNN_hbm_traffic.json (bench.py copies them into roofline.*.traffic).

    make_traffic_json.py SIM WORLDS fetch.db write.db >> entries

bytes = 2 * FETCH_SIZE KiB (gfx950 reports half of the fetched bytes) +
WRITE_SIZE KiB.  Launches per step = dispatches of the kernel / dispatches of
the end-of-replay health kernel (one per replay)."""
import json
import re
import sqlite3
import sys
from collections import defaultdict

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from summarize_pmc import short_name  # noqa: E402

# rocprof kernel (function) name -> the name bench.py's kernel table uses
BENCH_NAMES = [
    (r"sortHistogram", "SortArchetype:sort.histogram"),
    (r"sortOnesweep", "SortArchetype:sort.onesweep"),
    (r"sortCompactPrepare", "SortArchetype:sort.compact.prepare"),
    (r"sortCompactScatter", "SortArchetype:sort.compact.scatter"),
    (r"sortGather", "SortArchetype:sort.gather"),
    (r"sortFinalize", "SortArchetype:sort.finalize"),
    (r"sortSmall", "SortArchetype:sort.small"),
    (r"physicsStepLdsKernel", "physics:worldStep(LDS)"),
    # (main kernel of worlds beyond 128 bodies, and the fallback launch behind
    # the LDS kernel)
    (r"physicsStepKernel", "physics:worldStep(HBM)"),
    (r"bvhRefreshKernel", "physics:bvhRefresh"),
    # (every launch that several ParallelFor nodes share)
    (r"pforGroupKernel", "group[nodes sharing a launch]"),
    (r"physicsOrderKernel", "physics:orderWorlds"),
    (r"inputRingKernel", "input:ring"),
    (r"bvhUpdateKernel", "physics:bvhUpdate"),
    (r"updateLeafAndRefitEntry", "madrona::phys::broadphase::updateLeafAndRefitEntry"),
    (r"renderRaycast", "render:raycast"),
    (r"renderTlasBuild", "render:tlas.build"),
    (r"miscOpsKernel", "misc:clear/reset"),
    (r"statsKernel", "stats:health"),
]


def per_kernel(db_path, counter):
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    acc = defaultdict(lambda: [0.0, 0])
    for kname, cname, value in db.execute(
            f"select {name_col}, counter_name, value from counters_collection"):
        if cname != counter:
            continue
        a = acc[short_name(kname)]
        a[0] += value
        a[1] += 1
    return acc


def bench_name(short):
    for pat, name in BENCH_NAMES:
        if re.search(pat, short):
            return name
    m = re.match(r"parallelForKernel<(chain\[[^\]]+\])>", short) or \
        re.match(r"parallelForKernel<([\w:]+)>", short)
    return m.group(1) if m else short


def main():
    sim, worlds, fetch_db, write_db = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
    fetch = per_kernel(fetch_db, "FETCH_SIZE")
    write = per_kernel(write_db, "WRITE_SIZE")
    # Each pass is normalised by ITS OWN replay count: the bench repeats its
    # timed window until it is long enough, so the two passes (different counter,
    # different slowdown) replay the step graph a different number of times.
    # (Round 3 divided both by the fetch pass's count: at 65536 Escape-Room
    # worlds -- 262 replays in the fetch pass, 397 in the write pass -- every
    # WRITE_SIZE per step came out 1.52 x too large, the "write amplification"
    # of the gather in r03_hbm_traffic.json.)
    fetch_steps = max(n for k, (_, n) in fetch.items() if "statsKernel" in k)
    write_steps = max(n for k, (_, n) in write.items() if "statsKernel" in k)
    merged = defaultdict(lambda: [0.0, 0.0, 0])
    for short, (total, n) in fetch.items():
        merged[bench_name(short)][0] += total
        merged[bench_name(short)][2] += n
    for short, (total, _) in write.items():
        merged[bench_name(short)][1] += total
    entries = []
    for name, (f_kib, w_kib, n) in sorted(merged.items()):
        if "benchWindowMarker" in name or "gateKernel" in name:
            continue
        per_step = (2.0 * f_kib / fetch_steps + w_kib / write_steps) * 1024.0
        entries.append({"sim": sim, "worlds": worlds, "kernel": name,
                        "fetch_size_kib_per_step": round(f_kib / fetch_steps, 1),
                        "write_size_kib_per_step": round(w_kib / write_steps, 1),
                        "launches_per_step": round(n / fetch_steps, 2),
                        "replays_fetch_pass": fetch_steps,
                        "replays_write_pass": write_steps,
                        "traffic_bytes": int(per_step)})
    entries.append(step_entry(sim, worlds, entries))
    print(json.dumps(entries))


def in_step(entry):
    """Kernels of the step graph: launched about once per replay or more, and not
    the process's other GPU work (the bench's torch copy / triad bandwidth probe,
    its action ring set-up, hipMemcpy / hipMemset kernels, world construction)."""
    name = entry["kernel"]
    return (entry["launches_per_step"] >= 0.5 and
            not name.startswith(("at::native", "__amd_rocclr", "step:")) and
            "initWorlds" not in name)


def step_entry(sim, worlds, entries):
    mine = [e for e in entries if in_step(e)]
    return {"sim": sim, "worlds": worlds, "kernel": "step:all-kernels",
            "launches_per_step": round(sum(e["launches_per_step"] for e in mine), 2),
            "traffic_bytes": int(sum(e["traffic_bytes"] for e in mine))}


if __name__ == "__main__":
    main()
