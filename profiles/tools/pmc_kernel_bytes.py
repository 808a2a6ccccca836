#!/usr/bin/env python3
"""HBM traffic per LAUNCH of the kernels matching a pattern, from two rocprofv3
PMC passes of the same command (FETCH_SIZE, WRITE_SIZE; one counter per pass):

    pmc_kernel_bytes.py fetch.db write.db PATTERN SIM WORLDS NAME

prints one entry for profiles/rNN_hbm_traffic.json.  bytes = 2 * FETCH_SIZE KiB
(gfx950 reports half of the fetched bytes, MI355X_MICROARCH.md) + WRITE_SIZE KiB."""
import json
import sqlite3
import sys


def per_launch(db_path, counter, pattern):
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    total, n = 0.0, 0
    for kname, cname, value in db.execute(
            f"select {name_col}, counter_name, value from counters_collection"):
        if cname == counter and pattern in kname:
            total += value
            n += 1
    return total / max(n, 1), n


def main():
    fetch_db, write_db, pattern, sim, worlds, name = sys.argv[1:7]
    f_kib, n = per_launch(fetch_db, "FETCH_SIZE", pattern)
    w_kib, _ = per_launch(write_db, "WRITE_SIZE", pattern)
    print(json.dumps({"sim": sim, "worlds": int(worlds), "kernel": name,
                      "fetch_size_kib_per_launch": round(f_kib, 1),
                      "write_size_kib_per_launch": round(w_kib, 1),
                      "launches_measured": n,
                      "traffic_bytes": int((2.0 * f_kib + w_kib) * 1024.0)}))


if __name__ == "__main__":
    main()
