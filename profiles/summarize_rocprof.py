#!/usr/bin/env python3
"""Turns a rocprofv3 rocpd database (rocprofv3 --kernel-trace --stats -d DIR -o NAME)
into the per-kernel summary committed under profiles/ (CSV + aligned text)."""
import csv
import re
import sqlite3
import sys


def short(name):
    m = re.search(r"parallelForKernelIN\d+(\w+?)6EngineETnDaXadL_ZNS\d_\d+(\w+?System)", name)
    if m:
        return f"parallelForKernel<{m.group(1)}::{m.group(2)}>"
    name = re.sub(r"madrona::mwhip::\(anonymous namespace\)::", "mwhip::", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"\(.*", "", name)[:80]


def main():
    db_path, out_prefix = sys.argv[1], sys.argv[2]
    db = sqlite3.connect(db_path)
    # bench.py brackets its timed window with benchWindowMarker dispatches
    # (mwhip_mark_window): keep only what ran between the first two of them
    marks = [r[0] for r in db.execute(
        "select start from kernels where name like '%benchWindowMarker%' order by start")]
    where, window = "", "whole trace (no window markers)"
    if len(marks) >= 2:
        where = f" where start > {marks[0]} and start < {marks[1]}"
        window = (f"timed window only: {(marks[1] - marks[0]) / 1e6:.3f} ms between "
                  f"the benchWindowMarker dispatches")
    rows = db.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), "
        "max(end-start), max(grid_x), max(workgroup_x), max(vgpr_count), max(sgpr_count), "
        f"max(lds_size) from kernels{where} group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    with open(out_prefix + ".csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct",
                    "grid_x", "workgroup_x", "vgpr", "sgpr", "lds_bytes", "mangled"])
        for r in rows:
            w.writerow([short(r[0]), r[1], int(r[2]), round(r[3], 1), int(r[4]), int(r[5]),
                        round(100 * r[2] / total, 2), r[6], r[7], r[8], r[9], r[10], r[0]])
    with open(out_prefix + ".txt", "w") as f:
        f.write(f"# {window}\n")
        f.write(f"{'calls':>7} {'total_ms':>10} {'avg_us':>9} {'min_us':>8} {'max_us':>9} "
                f"{'pct':>6}  kernel\n")
        for r in rows:
            f.write(f"{r[1]:7d} {r[2] / 1e6:10.3f} {r[3] / 1e3:9.2f} {r[4] / 1e3:8.2f} "
                    f"{r[5] / 1e3:9.2f} {100 * r[2] / total:6.2f}  {short(r[0])}\n")
    print(open(out_prefix + ".txt").read())


if __name__ == "__main__":
    main()
