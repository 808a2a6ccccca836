#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters from a rocpd database
(rocprofv3 --kernel-trace --pmc ... -d DIR -o NAME)."""
import re
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    pattern = sys.argv[2] if len(sys.argv) > 2 else ""
    tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    view = next((t for t in tables if t == "counters_collection"), None)
    if view is None:
        print("tables:", tables)
        return
    cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    rows = db.execute(f"select {name_col}, counter_name, value from {view}").fetchall()
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for kname, cname, value in rows:
        if pattern and not re.search(pattern, kname):
            continue
        short = re.sub(r"\(.*", "", kname)[:70]
        acc[short][cname][0] += value
        acc[short][cname][1] += 1
    for kname, counters in acc.items():
        print(kname)
        for cname, (total, n) in sorted(counters.items()):
            print(f"    {cname:28s} avg/dispatch {total / n:16.1f}   (n={n})")


if __name__ == "__main__":
    main()
