"""Differential-parity helpers: step a simulator on the reference CPU backend
(oracle/_ref) and on the HIP backend with identical seeds and diff every dumped
column."""
from __future__ import annotations

import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from madrona_amd.simlib import Simulator, hip_lib_path, ref_lib_path  # noqa: E402


def compare_columns(ref_dump, hip_dump, rtol=1e-5):
    """Returns a list of human-readable mismatch strings (empty == parity)."""
    problems = []
    for name, (ref_rows, ref_counts) in ref_dump.items():
        hip_rows, hip_counts = hip_dump[name]
        if not np.array_equal(ref_counts, hip_counts):
            bad = np.nonzero(ref_counts != hip_counts)[0]
            problems.append(
                f"{name}: rows/world differ in {len(bad)} worlds, first w={bad[0]} "
                f"ref={ref_counts[bad[0]]} hip={hip_counts[bad[0]]}")
            continue
        if ref_rows.shape != hip_rows.shape:
            problems.append(f"{name}: shape {ref_rows.shape} vs {hip_rows.shape}")
            continue
        if not np.array_equal(ref_rows, hip_rows):
            bad = np.nonzero((ref_rows != hip_rows).any(axis=1))[0]
            problems.append(
                f"{name}: {len(bad)}/{len(ref_rows)} rows differ bitwise, first row "
                f"{bad[0]}: ref={ref_rows[bad[0]].tobytes().hex()} "
                f"hip={hip_rows[bad[0]].tobytes().hex()}")
    return problems


def float_close(ref_rows, hip_rows, rtol=1e-5):
    a = ref_rows.view(np.float32).astype(np.float64)
    b = hip_rows.view(np.float32).astype(np.float64)
    tol = rtol * np.maximum(1.0, np.maximum(np.abs(a), np.abs(b)))
    return bool((np.abs(a - b) <= tol).all())


def run_pair(sim, num_worlds, steps, seed=5, check_every=1, actions=None,
             check_init=True, ref_workers=1, hip_sim=None, **kw):
    """Steps both backends in lock step; returns (first mismatch list, step).
    ref_workers: threads of the reference CPU backend (0 = every core; worlds
    are independent, so the result does not depend on it).  hip_sim: another
    HIP build of the same simulator (e.g. "<sim>_portable")."""
    with Simulator(ref_lib_path(sim), num_worlds, seed=seed,
                   num_workers=ref_workers, **kw) as ref, \
            Simulator(hip_lib_path(hip_sim or sim), num_worlds, seed=seed,
                      **kw) as hip:
        if check_init:
            probs = compare_columns(ref.dump_all(), hip.dump_all())
            if probs:
                return probs, 0
        for s in range(1, steps + 1):
            if actions is not None:
                actions(ref, hip, s)
            ref.step(1)
            hip.step(1)
            if s % check_every == 0 or s == steps:
                probs = compare_columns(ref.dump_all(), hip.dump_all())
                if probs:
                    return probs, s
        # exported tensors must agree too
        for name in ref.tensor_names:
            if not np.array_equal(ref.read_tensor(name).view(np.uint8),
                                  hip.read_tensor(name).view(np.uint8)):
                return [f"tensor {name} differs"], steps
    return [], steps
