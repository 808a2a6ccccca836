#!/usr/bin/env python3
"""Per-kernel table of a step: declared (algorithmic) bytes next to the HBM
traffic the PMC passes measured (profiles/rNN_hbm_traffic.json), so that a sum
such as "ParallelFor traffic = 1.9 x algorithmic" names its kernels.

    python profiles/tools/traffic_table.py profiles/r06_bench_default_detail.json \
        profiles/r06_hbm_traffic.json escape_room_phys 8192 > profiles/r06_traffic_by_kernel.md
"""
import json
import sys


def main():
    detail, traffic, sim, worlds = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    kernels = json.load(open(detail))["kernels"]
    ents = [e for e in json.load(open(traffic))["entries"]
            if e.get("sim") == sim and e.get("worlds") == worlds]
    by_name = {}
    for e in ents:
        by_name.setdefault(e["kernel"], e)
    # kernels that appear several times in a step: the traffic entry is per STEP
    # (all launches of that name), the detail table has one row per launch
    print(f"| kernel ({sim}, {worlds} worlds) | launches / step | µs (sum) | declared MB | "
          f"PMC MB | PMC / declared | GB/s of PMC bytes |")
    print("|---|---|---|---|---|---|---|")
    def traffic_entry(name):
        """(entry, launches the entry covers) for a kernel of the detail table"""
        if name.startswith("group["):
            return by_name.get("group[nodes sharing a launch]")
        if name.startswith("physics:bvhRefresh"):
            return by_name.get("physics:bvhRefresh")     # (all three launches)
        if name == "physics:worldStep(fallback)":
            return by_name.get("physics:worldStep(HBM)")
        return by_name.get(name)

    # one row per traffic entry (the PMC passes cannot tell the launches of one
    # kernel apart): the three leaf refreshes of a step are one row
    groups = {}
    order = []
    for k in kernels:
        e = traffic_entry(k["name"])
        key = e["kernel"] if e else k["name"]
        if key not in groups:
            groups[key] = {"names": [], "us": 0.0, "algo": 0.0, "entry": e, "n": 0}
            order.append(key)
        g = groups[key]
        if k["name"] not in g["names"]:
            g["names"].append(k["name"])
        g["us"] += k["avg_us"]
        g["algo"] += (k.get("algo_MB") or k.get("algo_MB_signature_upper_bound") or 0.0)
        g["n"] += 1
    tot_algo = tot_pmc = tot_us = 0.0
    for key in order:
        g = groups[key]
        e = g["entry"]
        pmc = e["traffic_bytes"] / 1e6 if e else None
        algo, us = g["algo"], g["us"]
        ratio = f"{pmc / algo:.2f}" if pmc is not None and algo > 0 else "—"
        rate = f"{pmc / us * 1e3:.0f}" if pmc is not None and us > 0 else "—"
        label = " + ".join(n[:70] for n in g["names"])
        pmc_s = f"{pmc:.2f}" if pmc is not None else "—"
        print(f"| `{label}` | {g['n']} | {us:.1f} | {algo:.2f} | {pmc_s} | {ratio} | {rate} |")
        tot_us += us
        tot_algo += algo
        tot_pmc += pmc or 0.0
    print(f"| **step** | {len(kernels)} | {tot_us:.1f} | {tot_algo:.1f} | {tot_pmc:.1f} | "
          f"{tot_pmc / max(tot_algo, 1e-9):.2f} | {tot_pmc / tot_us * 1e3:.0f} |")


if __name__ == "__main__":
    main()
