#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters from a rocpd database
(rocprofv3 --kernel-trace --pmc ... -d DIR -o NAME)."""
import re
import sqlite3
import subprocess
import sys
from collections import defaultdict


_demangled = {}


def short_name(kname):
    """Readable kernel name: demangled, anonymous-namespace noise and the
    argument list dropped."""
    if kname not in _demangled:
        name = kname
        if name.startswith("_Z"):
            try:
                name = subprocess.run(["c++filt", name], capture_output=True,
                                      text=True).stdout.strip() or name
            except OSError:
                pass
        if name.startswith("_Z"):
            # c++filt of this image does not know `TnDa` (auto non-type
            # template parameter): pull the system name out by hand
            m = re.search(r"parallelForKernelIN\d+(\w+?)\d+EngineE.*?XadL_ZNS\d_(\d+)", name)
            chain = re.search(r"parallelForKernelIN\d+(\w+?)\d+EngineE.*?8RowChainI"
                              r"XadL_ZNS\d_(\d+)", name)
            if chain:
                # mwhip::RowChain<&ns::fnA, &ns::fnB, ...>::run: both systems' names
                n = int(chain.group(2))
                first = name[chain.end():chain.end() + n]
                second = re.search(r"XadL_ZNS\d_(\d+)", name[chain.end() + n:])
                at = chain.end() + n + (second.end() if second else 0)
                second = name[at:at + int(second.group(1))] if second else "?"
                ns = chain.group(1)
                name = (f"parallelForKernel<chain[{ns}::{first} > {ns}::{second}]>(")
            elif m:
                n = int(m.group(2))
                name = ("parallelForKernel<" + m.group(1) + "::" +
                        name[m.end():m.end() + n] + ">(")
        name = name.replace("(anonymous namespace)::", "")
        name = re.sub(r"^void ", "", name)
        # parallelForKernel<Ctx, &fn, ...>: keep the system's name
        m = re.match(r".*parallelForKernel<[^,]+, &?\(?([\w:]+)", name)
        if m:
            name = "parallelForKernel<" + m.group(1) + ">"
        _demangled[kname] = re.sub(r"\(.*", "", name)[:90]
    return _demangled[kname]


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
        short = short_name(kname)
        acc[short][cname][0] += value
        acc[short][cname][1] += 1
    for kname, counters in acc.items():
        print(kname)
        for cname, (total, n) in sorted(counters.items()):
            print(f"    {cname:28s} avg/dispatch {total / n:16.1f}   (n={n})")


if __name__ == "__main__":
    main()
