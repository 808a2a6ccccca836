"""In-tree builds (no network, no pip): everything is a Makefile over hipcc/clang."""
from __future__ import annotations

import os
import subprocess

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE_ROOT = "/root/reference"


def _make(directory: str, *targets: str) -> None:
    cmd = ["make", "-C", directory, "-j", str(min(os.cpu_count() or 4, 16)), *targets]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(
            f"{' '.join(cmd)} failed:\n{res.stdout[-4000:]}\n{res.stderr[-8000:]}")


def build_hip() -> None:
    """libmadrona_hip.so + every simulator in sims/ for gfx950."""
    _make(os.path.join(REPO_ROOT, "madrona_amd"), "all")
    # the MADRONA_TRACING flavour (device event log in every kernel,
    # madrona/mw_gpu/tracing.hpp) of the runtime and of the two simulators its
    # test runs, apart from the product build
    _make(os.path.join(REPO_ROOT, "madrona_amd"), "TRACING=1", "OUT=_build_tracing",
          "runtime", "_build_tracing/libescape_room_hip.so",
          "_build_tracing/libcartpole_hip.so")


def build_oracle() -> None:
    """oracle/restate (plain C) always; oracle/_ref (the reference CPU backend
    compiled from /root/reference) only where the reference tree exists."""
    oracle_dir = os.path.join(REPO_ROOT, "oracle")
    _make(oracle_dir, "restate")
    if os.path.isdir(os.path.join(REFERENCE_ROOT, "src")):
        _make(oracle_dir, "parity", "speed")
