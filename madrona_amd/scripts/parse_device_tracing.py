#!/usr/bin/env python3
"""Reads the device event log of a MADRONA_TRACING build
(<dir>/<pid or $MADRONA_MWGPU_TRACE_NAME>_madrona_device_tracing.bin, 40-byte
records: madrona_amd/include/madrona/mw_gpu/tracing.hpp) and prints, per step,
what the reference's scripts/parse_device_tracing.py tabulates per node --
start, duration, share of the step -- plus what one kernel per node makes
available: workgroups, compute units used, how evenly they finished.

    python madrona_amd/scripts/parse_device_tracing.py /tmp/1234_madrona_device_tracing.bin
        [--nodes /tmp/1234_madrona_device_tracing_nodes.bin] [--step N] [--json]

The record layout and event codes are the reference's (DeviceLog,
DeviceEvent), so its parser reads the same file; it draws megakernel block
timelines (PIL) this backend has no use for."""
import argparse
import json
import os
import sys

import numpy as np

RECORD = np.dtype([("event", "<u4"), ("funcID", "<u4"), ("numInvocations", "<u4"),
                   ("nodeID", "<u4"), ("warpID", "<u4"), ("blockID", "<u4"),
                   ("smID", "<u4"), ("logIndex", "<u4"), ("cycleCount", "<u8")])
assert RECORD.itemsize == 40
CALIBRATION, NODE_START, NODE_FINISH, BLOCK_START, BLOCK_WAIT, BLOCK_EXIT = range(6)


def read_log(path):
    raw = np.fromfile(path, dtype=np.uint8)
    if len(raw) % RECORD.itemsize:
        raise ValueError(f"{path}: {len(raw)} bytes is not a whole number of records")
    return raw.view(RECORD)


def split_steps(log):
    """A step starts with the calibration record that took slot 0 of its log."""
    starts = np.flatnonzero((log["event"] == CALIBRATION) & (log["logIndex"] == 0))
    return [log[a:b] for a, b in zip(starts, list(starts[1:]) + [len(log)])]


def analyse_step(step, names):
    t0 = int(step["cycleCount"][step["event"] == CALIBRATION][0])
    exits = step["cycleCount"][step["event"] == BLOCK_EXIT]
    total = int(exits.max()) - t0 if len(exits) else int(step["cycleCount"].max()) - t0
    nodes = []
    starts = step[step["event"] == NODE_START]
    finish = {int(r["nodeID"]): r for r in step[step["event"] == NODE_FINISH]}
    bs = step[step["event"] == BLOCK_START]
    bw = step[step["event"] == BLOCK_WAIT]
    for r in starts[np.argsort(starts["nodeID"], kind="stable")]:
        nid = int(r["nodeID"])
        s = bs[bs["nodeID"] == nid]
        w = bw[bw["nodeID"] == nid]
        # a workgroup = (blockID, numInvocations): its first and last instruction
        first = int(s["cycleCount"].min()) if len(s) else int(r["cycleCount"])
        last = int(finish[nid]["cycleCount"]) if nid in finish else first
        per_wg = None
        if len(s) and len(s) == len(w):
            so = s[np.argsort(s["numInvocations"], kind="stable")]
            wo = w[np.argsort(w["numInvocations"], kind="stable")]
            per_wg = (wo["cycleCount"].astype(np.int64) - so["cycleCount"].astype(np.int64))
        fid = int(r["funcID"])
        nodes.append({
            "nodeID": nid,
            "funcID": fid,
            "name": names[fid] if fid < len(names) else f"func {fid}",
            "threads": int(r["numInvocations"]),
            "marked_ns": int(r["cycleCount"]) - t0,
            "start_ns": first - t0,
            "duration_ns": last - first,
            "workgroups": int(len(s)),
            "compute_units": int(len(np.unique(s["smID"]))) if len(s) else 0,
            "workgroup_ns_mean": float(per_wg.mean()) if per_wg is not None else None,
            "workgroup_ns_max": int(per_wg.max()) if per_wg is not None else None,
        })
    busy = sum(n["duration_ns"] for n in nodes)
    for n in nodes:
        n["percent_of_kernels"] = 100.0 * n["duration_ns"] / busy if busy else 0.0
    calib = step[step["event"] == CALIBRATION][0]
    return {"total_ns": total, "kernel_ns": busy, "nodes": nodes,
            "waves_per_workgroup": int(calib["funcID"]),
            "compute_units": int(calib["nodeID"]), "records": int(len(step))}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__,
                                 formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("log")
    ap.add_argument("--nodes", help="kernel names, one per line (funcID = line number)")
    ap.add_argument("--step", type=int, default=-1, help="which step to print (default: last)")
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--node", type=int, default=None,
                    help="also print the workgroup statistics of this node")
    args = ap.parse_args(argv)

    nodes_path = args.nodes
    if nodes_path is None:
        guess = args.log.replace("_madrona_device_tracing.bin",
                                 "_madrona_device_tracing_nodes.bin")
        nodes_path = guess if guess != args.log and os.path.exists(guess) else None
    names = open(nodes_path).read().splitlines() if nodes_path else []

    steps = split_steps(read_log(args.log))
    if not steps:
        print("no complete step in the log", file=sys.stderr)
        return 1
    result = analyse_step(steps[args.step], names)
    result["steps_in_log"] = len(steps)
    if args.json:
        print(json.dumps(result))
        return 0
    print(f"{len(steps)} steps in the log; step {args.step % len(steps)}: "
          f"{result['total_ns'] / 1e3:.1f} us, {result['kernel_ns'] / 1e3:.1f} us inside kernels, "
          f"{result['records']} records")
    print(f"{'node':>4} {'start us':>9} {'dur us':>8} {'%':>5} {'wgs':>6} {'CUs':>4} "
          f"{'wg mean us':>10} {'wg max us':>9}  kernel")
    for n in result["nodes"]:
        mean = f"{n['workgroup_ns_mean'] / 1e3:10.2f}" if n["workgroup_ns_mean"] is not None else " " * 10
        mx = f"{n['workgroup_ns_max'] / 1e3:9.2f}" if n["workgroup_ns_max"] is not None else " " * 9
        print(f"{n['nodeID']:4d} {n['start_ns'] / 1e3:9.2f} {n['duration_ns'] / 1e3:8.2f} "
              f"{n['percent_of_kernels']:5.1f} {n['workgroups']:6d} {n['compute_units']:4d} "
              f"{mean} {mx}  {n['name']}")
    if args.node is not None:
        print_node(steps[args.step], args.node)
    return 0


def print_node(step, nid):
    """How a kernel's workgroups spread over its duration: the quantiles of their
    own run times, and when which share of them had started / finished."""
    s = step[(step["event"] == BLOCK_START) & (step["nodeID"] == nid)]
    w = step[(step["event"] == BLOCK_WAIT) & (step["nodeID"] == nid)]
    if not len(s) or len(s) != len(w):
        print(f"node {nid}: no complete workgroup records")
        return
    s = s[np.argsort(s["numInvocations"], kind="stable")]
    w = w[np.argsort(w["numInvocations"], kind="stable")]
    t0 = int(s["cycleCount"].min())
    begin = s["cycleCount"].astype(np.int64) - t0
    end = w["cycleCount"].astype(np.int64) - t0
    run = end - begin
    q = [0, 10, 50, 90, 99, 100]
    print(f"node {nid}: {len(s)} workgroups, kernel {end.max() / 1e3:.1f} us")
    print("  run time us, percentiles " + str(q) + ": " +
          " ".join(f"{np.percentile(run, p) / 1e3:.1f}" for p in q))
    print("  started by us, percentiles:  " +
          " ".join(f"{np.percentile(begin, p) / 1e3:.1f}" for p in q))
    print("  finished by us, percentiles: " +
          " ".join(f"{np.percentile(end, p) / 1e3:.1f}" for p in q))
    # workgroups in flight over time (20 bins)
    edges = np.linspace(0, end.max(), 21)
    mids = 0.5 * (edges[1:] + edges[:-1])
    flight = [(int(((begin <= m) & (end > m)).sum())) for m in mids]
    print("  in flight at 2.5 %, 7.5 %, ... of the kernel: " + " ".join(map(str, flight)))
    # dispatch order: mean run time of each tenth of the workgroups, by index
    tenths = np.array_split(run, 10)
    print("  mean run time us by tenth of the workgroup index: " +
          " ".join(f"{t.mean() / 1e3:.1f}" for t in tenths if len(t)))


if __name__ == "__main__":
    sys.exit(main())
