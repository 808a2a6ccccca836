#!/usr/bin/env python3
"""trace_wgs.py LOG NODE: per compute unit and per XCD, how the workgroups of one
kernel of a MADRONA_TRACING log filled their slots (last step of the log)."""
import sys
import numpy as np
sys.path.insert(0, "madrona_amd/scripts")
from parse_device_tracing import read_log, split_steps, BLOCK_START, BLOCK_WAIT

log, nid = sys.argv[1], int(sys.argv[2])
for which in (-1, -3):
    step = split_steps(read_log(log))[which]
    s = step[(step["event"] == BLOCK_START) & (step["nodeID"] == nid)]
    w = step[(step["event"] == BLOCK_WAIT) & (step["nodeID"] == nid)]
    s = s[np.argsort(s["numInvocations"], kind="stable")]
    w = w[np.argsort(w["numInvocations"], kind="stable")]
    t0 = int(s["cycleCount"].min())
    begin = (s["cycleCount"].astype(np.int64) - t0) / 1e3
    end = (w["cycleCount"].astype(np.int64) - t0) / 1e3
    run = end - begin
    cu = s["smID"].astype(np.int64)
    wg = s["numInvocations"].astype(np.int64)
    print(f"step {which}: {len(s)} workgroups, kernel {end.max():.1f} us, "
          f"sum of run times {run.sum() / 1e3:.1f} ms = {run.sum() / end.max():.0f} in flight on average")
    # slots per CU
    per_cu = {}
    for c in np.unique(cu):
        m = cu == c
        b, e = begin[m], end[m]
        peak = max(int(((b <= t) & (e > t)).sum()) for t in b)
        per_cu[int(c)] = (int(m.sum()), peak, float(e.max()), float(run[m].sum()))
    counts = np.array([v[0] for v in per_cu.values()])
    peaks = np.array([v[1] for v in per_cu.values()])
    lasts = np.array([v[2] for v in per_cu.values()])
    busy = np.array([v[3] for v in per_cu.values()])
    print(f"  compute units {len(per_cu)}; workgroups per CU min/mean/max {counts.min()}/{counts.mean():.1f}/{counts.max()}; "
          f"peak in flight per CU min/mean/max {peaks.min()}/{peaks.mean():.1f}/{peaks.max()}")
    print(f"  last finish per CU us: min {lasts.min():.1f} p10 {np.percentile(lasts, 10):.1f} "
          f"p50 {np.percentile(lasts, 50):.1f} p90 {np.percentile(lasts, 90):.1f} max {lasts.max():.1f}")
    print(f"  sum of run times per CU us: min {busy.min():.0f} p50 {np.percentile(busy, 50):.0f} max {busy.max():.0f}")
    # by dispatch order (workgroup index): start, run
    order = np.argsort(wg)
    for name, arr in (("start", begin), ("run", run), ("end", end)):
        parts = np.array_split(arr[order], 16)
        print(f"  {name:5s} us by sixteenth of the workgroup index: " + " ".join(f"{p.mean():.0f}" for p in parts))
    parts = np.array_split(run[order], 16)
    print("  run max by sixteenth: " + " ".join(f"{p.max():.0f}" for p in parts))
    # the twenty that finish last
    lastk = np.argsort(end)[-12:]
    print("  the last twelve to finish (index, cu, start, run, end): " +
          "; ".join(f"{wg[i]} {cu[i]} {begin[i]:.0f} {run[i]:.0f} {end[i]:.0f}" for i in lastk))
    # run time vs how many neighbours on the CU
    edges = np.linspace(0, end.max(), 25)
    mids = 0.5 * (edges[1:] + edges[:-1])
    print("  in flight over time: " + " ".join(str(int(((begin <= m) & (end > m)).sum())) for m in mids))
