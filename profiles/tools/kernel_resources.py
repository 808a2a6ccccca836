#!/usr/bin/env python3
"""Registers / LDS / scratch of the kernels of a code object whose name matches
PATTERN (llvm-readelf --notes on the gfx950 image bundled in a .so).

    python profiles/tools/kernel_resources.py LIB.so PATTERN"""
import re
import subprocess
import sys
import tempfile
import os

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
BUNDLER = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"
KEYS = (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count",
        ".sgpr_spill_count", ".group_segment_fixed_size",
        ".private_segment_fixed_size", ".max_flat_workgroup_size")


def notes(lib):
    out = subprocess.run([READELF, "--notes", lib], capture_output=True, text=True).stdout
    if ".vgpr_count" in out:
        return out
    # host library: the device image sits in the .hip_fatbin section
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "-O", "binary",
                        "--only-section=.hip_fatbin", lib, fat], check=True)
        img = os.path.join(tmp, "gfx950.co")
        subprocess.run([BUNDLER, "--unbundle", "--type=o", f"--input={fat}",
                        f"--output={img}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], check=True)
        return subprocess.run([READELF, "--notes", img], capture_output=True,
                              text=True).stdout


def main():
    lib, pat = sys.argv[1], sys.argv[2]
    txt = notes(lib)
    for blk in re.split(r"\n\s+- \.", txt):
        m = re.search(r"\.name:\s+(\S+)", blk)
        if not m or not re.search(pat, m.group(1)):
            continue
        vals = {}
        for k in KEYS:
            mm = re.search(re.escape(k[1:]) + r":\s+(\d+)", blk)
            vals[k[1:]] = int(mm.group(1)) if mm else None
        print(m.group(1)[:90], vals)


if __name__ == "__main__":
    main()
