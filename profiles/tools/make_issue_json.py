#!/usr/bin/env python3
"""Instruction-issue counters per LAUNCH of the kernels that are bound by
instruction issue rather than by HBM (the fused physics step, the ray cast), from
rocprofv3 PMC passes of a bench command -> entries for
profiles/rNN_issue_counters.json (bench.py copies them into the `valu-issue`
rooflines next to the HBM ones).

    make_issue_json.py SIM WORLDS pass1.db [pass2.db ...] >> entries

Counters (MI355X_MICROARCH.md, SQ block): SQ_INSTS_VALU / SQ_INSTS_SALU /
SQ_INSTS_LDS = wave-instructions issued; SQ_WAVES = wavefronts launched;
SQ_WAVE_CYCLES / SQ_ACTIVE_INST_ANY / SQ_WAIT_ANY / SQ_WAIT_INST_ANY in
quad-cycles summed over waves (issuing / parked on s_waitcnt / issue stalls);
SQ_BUSY_CYCLES = cycles the SQs had work."""
import json
import re
import sqlite3
import sys
from collections import defaultdict

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from summarize_pmc import short_name  # noqa: E402

KERNELS = [
    (r"physicsStepLdsKernel", "physics:worldStep(LDS)"),
    (r"physicsStepKernel", "physics:worldStep(HBM)"),
    (r"bvhRefreshKernel", "physics:bvhRefresh"),
    (r"renderRaycast", "render:raycast"),
    (r"resetSystem", "resetSystem"),
    (r"lidarSystem", "lidarSystem"),
]


def main():
    sim, worlds = sys.argv[1], int(sys.argv[2])
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for path in sys.argv[3:]:
        db = sqlite3.connect(path)
        cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
        name_col = "kernel_name" if "kernel_name" in cols else "name"
        for kname, cname, value in db.execute(
                f"select {name_col}, counter_name, value from counters_collection"):
            short = short_name(kname)
            for pattern, bench_name in KERNELS:
                if re.search(pattern, short) or re.search(pattern, kname):
                    a = acc[bench_name][cname]
                    a[0] += value
                    a[1] += 1
                    break
    for bench_name, counters in acc.items():
        entry = {"sim": sim, "worlds": worlds, "kernel": bench_name,
                 "launches_measured": max(n for _, n in counters.values())}
        for cname, (total, n) in sorted(counters.items()):
            entry[cname] = round(total / max(n, 1), 1)
        print(json.dumps(entry))


if __name__ == "__main__":
    main()
