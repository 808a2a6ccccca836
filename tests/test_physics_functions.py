"""CPU-only, function-level physics parity: madrona_amd's physics headers are
compiled for the host (tests/shims/phys_host_shim.cpp -> libphys_host_test.so)
and diffed against the reference's own code (oracle/_ref/libphys_ref.so, built
from /root/reference by oracle/Makefile):

* the asset baker (half-edge mesh, Newell planes, AABBs, mass properties with
  the McAdams inertia diagonalisation) -- bit exact;
* the scalar narrowphase on random box-box / box-plane pairs (SAT face and edge
  queries, clipping, manifold reduction) -- bit exact, for both hull
  representations the device code uses (stored in scratch, and evaluated
  lazily from the object-space mesh).
"""
import ctypes as C
import os

import numpy as np
import pytest

from madrona_amd.simlib import HIP_BUILD_DIR, REF_BUILD_DIR

AMD = os.path.join(HIP_BUILD_DIR, "libphys_host_test.so")
REF = os.path.join(REF_BUILD_DIR, "libphys_ref.so")

CUBE_VERTS = np.array([
    [-.5, -.5, -.5], [.5, -.5, -.5], [.5, .5, -.5], [-.5, .5, -.5],
    [-.5, -.5, .5], [.5, -.5, .5], [.5, .5, .5], [-.5, .5, .5]], np.float32)
CUBE_IDX = np.array([0, 3, 2, 1, 4, 5, 6, 7, 0, 1, 5, 4,
                     2, 3, 7, 6, 0, 4, 7, 3, 1, 2, 6, 5], np.uint32)
CUBE_COUNTS = np.array([4] * 6, np.uint32)

# a wedge (triangular prism, 6 vertices, 2 triangles + 3 quads): mixed face
# sizes, non axis-aligned planes, off-centre mass
WEDGE_VERTS = np.array([
    [0, 0, 0], [2, 0, 0], [0, 1, 0], [0, 0, 1.5], [2, 0, 1.5], [0, 1, 1.5]],
    np.float32)
WEDGE_IDX = np.array([0, 2, 1, 3, 4, 5, 0, 1, 4, 3, 1, 2, 5, 4, 2, 0, 3, 5],
                     np.uint32)
WEDGE_COUNTS = np.array([3, 3, 4, 4, 4], np.uint32)



def _prism(n, r, h):
    """Regular n-gon prism (sims/ball_pit's "coins"): two n-vertex caps."""
    ang = 2 * np.pi * np.arange(n) / n
    ring = np.stack([r * np.cos(ang), r * np.sin(ang)], 1)
    verts = np.concatenate([np.c_[ring, np.full(n, -h / 2)],
                            np.c_[ring, np.full(n, h / 2)]]).astype(np.float32)
    idx = [0] + list(range(n - 1, 0, -1)) + [n + i for i in range(n)]
    counts = [n, n]
    for i in range(n):
        j = (i + 1) % n
        idx += [i, j, n + j, n + i]
        counts.append(4)
    return verts, np.array(idx, np.uint32), np.array(counts, np.uint32)


MESHES = {"cube": (CUBE_VERTS, CUBE_IDX, CUBE_COUNTS),
          "wedge": (WEDGE_VERTS, WEDGE_IDX, WEDGE_COUNTS),
          "coin16": _prism(16, 1.0, 0.5)}


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


@pytest.fixture(scope="module")
def libs(built):
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref not built here (no /root/reference)")
    amd, ref = C.CDLL(AMD), C.CDLL(REF)
    bake_args = [C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint32),
                 C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_int32),
                 C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32,
                 C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32]
    for fn in (amd.amd_bake_objects, ref.ref_bake_objects):
        fn.restype = C.c_int32
        fn.argtypes = bake_args
    pair_args = [C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint32),
                 C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_float),
                 C.POINTER(C.c_float), C.c_int32]
    ref.ref_collide_pair.argtypes = pair_args + [C.POINTER(C.c_float)]
    amd.amd_collide_pair.argtypes = pair_args + [C.c_int32, C.POINTER(C.c_float)]
    return amd, ref


@pytest.mark.parametrize("mesh", ["cube", "wedge", "coin16"])
def test_asset_baker_matches_reference(libs, mesh):
    amd, ref = libs
    verts, idx, counts = MESHES[mesh]
    types = np.array([0, 0, 1, 2, 0], np.int32)
    inv_mass = np.array([0.075, 1.0, 0.0, 0.5, 0.0], np.float32)
    radius = np.array([0, 0, 0, 0.7, 0], np.float32)

    outs = []
    for fn in (ref.ref_bake_objects, amd.amd_bake_objects):
        floats = np.zeros(1024, np.float32)
        hedges = np.zeros(1024, np.uint32)
        n = fn(_ptr(verts, C.c_float), len(verts), _ptr(idx, C.c_uint32),
               _ptr(counts, C.c_uint32), len(counts), _ptr(types, C.c_int32),
               _ptr(inv_mass, C.c_float), _ptr(radius, C.c_float), len(types),
               _ptr(floats, C.c_float), len(floats), _ptr(hedges, C.c_uint32),
               len(hedges))
        assert 0 < n <= len(floats)
        outs.append((n, floats.copy(), hedges.copy()))
    (n_ref, f_ref, h_ref), (n_amd, f_amd, h_amd) = outs
    assert n_ref == n_amd
    assert np.array_equal(h_ref, h_amd), "half-edge connectivity differs"
    # bit-exact floats (inf for the plane's inertia compares equal by bits)
    assert np.array_equal(f_ref[:n_ref].view(np.uint32), f_amd[:n_amd].view(np.uint32))


def _random_quat(rng, tilt):
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    angle = rng.uniform(-tilt, tilt)
    return np.concatenate([[np.cos(angle / 2)], np.sin(angle / 2) * axis]).astype(np.float32)


@pytest.mark.parametrize("mesh,plane", [("cube", 0), ("cube", 1), ("wedge", 0),
                                        ("wedge", 1), ("cube", 2), ("wedge", 2),
                                        ("coin16", 0), ("coin16", 1), ("coin16", 2)])
def test_narrowphase_matches_reference(libs, mesh, plane):
    amd, ref = libs
    verts, idx, counts = MESHES[mesh]
    rng = np.random.default_rng(11 + plane)

    kinds = {"none": 0, "face_a": 0, "face_b": 0, "edge": 0, "plane": 0}
    for trial in range(1500):
        # mostly near-resting configurations (small tilts / overlaps), some wild
        tilt = 0.05 if trial % 3 else np.pi
        a = np.concatenate([rng.uniform(-1, 1, 3) * [1.5, 1.5, 0.3] + [0, 0, 0.7],
                            _random_quat(rng, tilt),
                            rng.uniform(0.6, 1.8, 3)]).astype(np.float32)
        if plane == 2:
            # a = sphere (uniform scale), b = hull; near, touching, inside, far
            r = rng.uniform(0.5, 1.5)
            a = np.concatenate([rng.uniform(-1, 1, 3) * [2.2, 2.2, 2.2],
                                [1, 0, 0, 0], [r, r, r]]).astype(np.float32)
            b = np.concatenate([rng.uniform(-1, 1, 3) * 0.3,
                                _random_quat(rng, np.pi),
                                rng.uniform(0.6, 2.5, 3)]).astype(np.float32)
        elif plane:
            b = np.array([0, 0, rng.uniform(-0.1, 0.3), 1, 0, 0, 0, 1, 1, 1], np.float32)
        else:
            b = np.concatenate([rng.uniform(-1, 1, 3) * [1.5, 1.5, 0.6],
                                _random_quat(rng, tilt),
                                rng.uniform(0.6, 1.8, 3)]).astype(np.float32)

        expect = np.zeros(28, np.float32)
        ref.ref_collide_pair(_ptr(verts, C.c_float), len(verts), _ptr(idx, C.c_uint32),
                             _ptr(counts, C.c_uint32), len(counts), _ptr(a, C.c_float),
                             _ptr(b, C.c_float), plane, _ptr(expect, C.c_float))
        for mode in (0, 1):
            got = np.zeros(28, np.float32)
            amd.amd_collide_pair(_ptr(verts, C.c_float), len(verts),
                                 _ptr(idx, C.c_uint32), _ptr(counts, C.c_uint32),
                                 len(counts), _ptr(a, C.c_float), _ptr(b, C.c_float),
                                 plane, mode, _ptr(got, C.c_float))
            # the reference leaves unused manifold slots uninitialised
            used = 6 + 4 * int(expect[2])
            assert np.array_equal(expect[:used].view(np.uint32),
                                  got[:used].view(np.uint32)), \
                (trial, mode, expect, got)

        if expect[0] == 0:
            kinds["none"] += 1
        elif plane == 2:
            kinds["sphere"] = kinds.get("sphere", 0) + 1
        elif plane:
            kinds["plane"] += 1
        elif expect[2] == 1:
            kinds["edge"] += 1
        else:
            kinds["face_a" if expect[1] else "face_b"] += 1

    # the sweep must actually exercise the feature types
    if plane == 2:
        assert kinds["sphere"] > 200 and kinds["none"] > 200, kinds
    elif plane:
        assert kinds["plane"] > 150 and kinds["none"] > 50, kinds
    else:
        assert kinds["face_a"] > 50 and kinds["face_b"] > 50, kinds
        assert kinds["edge"] > 20 and kinds["none"] > 100, kinds
