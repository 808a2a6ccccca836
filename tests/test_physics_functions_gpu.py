"""Function-level narrowphase parity ON THE DEVICE (``pytest -m gpu``): the three
code paths the step kernels take for a candidate pair -- hulls stored in private
scratch, LazyHull per lane with an LDS row, and the wave-cooperative hull-hull
SAT -- run pair by pair on the GPU (tests/shims/phys_device_shim.hip ->
libphys_device_test.so) and are diffed bit for bit against the reference's own
narrowphase (oracle/_ref/libphys_ref.so) on the same random configurations as
tests/test_physics_functions.py, for a cube, a wedge, and a 16-sided prism whose
caps outgrow every LDS staging buffer (so the fallbacks run)."""
import ctypes as C
import os

import numpy as np
import pytest

from madrona_amd.simlib import HIP_BUILD_DIR, REF_BUILD_DIR
from test_physics_functions import MESHES, _ptr, _random_quat

pytestmark = pytest.mark.gpu

DEV = os.path.join(HIP_BUILD_DIR, "libphys_device_test.so")
REF = os.path.join(REF_BUILD_DIR, "libphys_ref.so")


@pytest.fixture(scope="module")
def libs(built):
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref missing on this box")
    import torch  # noqa: F401  (torch's HIP runtime first, see simlib)
    C.CDLL(os.path.join(HIP_BUILD_DIR, "libmadrona_hip.so"), mode=C.RTLD_GLOBAL)
    dev, ref = C.CDLL(DEV), C.CDLL(REF)
    mesh_args = [C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint32),
                 C.POINTER(C.c_uint32), C.c_uint32]
    ref.ref_collide_pair.argtypes = mesh_args + [
        C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_float)]
    dev.dev_collide_pairs.restype = C.c_int32
    dev.dev_collide_pairs.argtypes = mesh_args + [
        C.POINTER(C.c_float), C.c_uint32, C.c_int32, C.c_int32,
        C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    return dev, ref


def _configs(rng, kind, n):
    out = np.zeros((n, 20), np.float32)
    for t in range(n):
        tilt = 0.05 if t % 3 else np.pi
        a = np.concatenate([rng.uniform(-1, 1, 3) * [1.5, 1.5, 0.3] + [0, 0, 0.7],
                            _random_quat(rng, tilt), rng.uniform(0.6, 1.8, 3)])
        if kind == 2:
            r = rng.uniform(0.5, 1.5)
            a = np.concatenate([rng.uniform(-1, 1, 3) * 2.2, [1, 0, 0, 0], [r, r, r]])
            b = np.concatenate([rng.uniform(-1, 1, 3) * 0.3, _random_quat(rng, np.pi),
                                rng.uniform(0.6, 2.5, 3)])
        elif kind == 1:
            b = np.array([0, 0, rng.uniform(-0.1, 0.3), 1, 0, 0, 0, 1, 1, 1])
        else:
            # stacked / touching / interpenetrating / apart
            b = np.concatenate([a[:3] + rng.uniform(-1, 1, 3) * [1.2, 1.2, 0.9],
                                _random_quat(rng, tilt), rng.uniform(0.6, 1.8, 3)])
        out[t, :10], out[t, 10:] = a, b
    return out


@pytest.mark.parametrize("mesh", ["cube", "wedge", "coin16"])
# (modes 3 / 4: the cooperative hull-hull test on 32- / 16-lane groups, two / four
# pairs per wavefront -- what the two-worlds-per-wavefront step kernel runs)
@pytest.mark.parametrize("kind,mode", [(0, 0), (0, 2), (0, 3), (0, 4), (1, 0), (1, 1),
                                       (2, 0), (2, 1)])
def test_device_narrowphase_matches_reference(libs, mesh, kind, mode):
    dev, ref = libs
    verts, idx, counts = MESHES[mesh]
    n = 1024
    pairs = _configs(np.random.default_rng(100 + 10 * kind + mode), kind, n)

    expect = np.zeros((n, 28), np.float32)
    for t in range(n):
        a, b = pairs[t, :10].copy(), pairs[t, 10:].copy()
        ref.ref_collide_pair(_ptr(verts, C.c_float), len(verts), _ptr(idx, C.c_uint32),
                             _ptr(counts, C.c_uint32), len(counts), _ptr(a, C.c_float),
                             _ptr(b, C.c_float), kind, _ptr(expect[t], C.c_float))

    got = np.zeros((n, 28), np.float32)
    flags = np.zeros(n, np.int32)
    rc = dev.dev_collide_pairs(_ptr(verts, C.c_float), len(verts), _ptr(idx, C.c_uint32),
                               _ptr(counts, C.c_uint32), len(counts),
                               _ptr(pairs, C.c_float), n, kind, mode,
                               _ptr(got, C.c_float), _ptr(flags, C.c_int32))
    assert rc == 0
    assert not (flags & 2).any(), "unsupported primitive pair"

    checked = hits = 0
    for t in range(n):
        if flags[t] & 1:
            continue    # face too big for the LDS scratch: the kernel reruns it as mode 0
        used = 6 + 4 * int(expect[t, 2])
        # the reference leaves unused manifold slots uninitialised
        assert np.array_equal(expect[t, :used].view(np.uint32),
                              got[t, :used].view(np.uint32)), \
            (t, pairs[t], expect[t, :used], got[t, :used])
        checked += 1
        hits += int(expect[t, 0] != 0)
    if mesh == "coin16" and mode != 0 and kind != 2:
        assert checked < n, "the big-face fallback never triggered"
    else:
        assert checked == n
    assert hits > n // 20, "the configurations hardly ever touch"


def test_reference_kats_on_device(built):
    """The reference's known answers for this path (tests/math.cpp:23-48
    quaternions, tests/gjk.cpp:19-48 GJK simplex solves), computed by the
    overlay's math.hpp / phys_impl/gjk.hpp on the MI355X, with the reference's
    own tolerances."""
    dev = C.CDLL(DEV)
    dev.dev_reference_kats.restype = C.c_int32
    dev.dev_reference_kats.argtypes = [C.POINTER(C.c_float)]
    out = (C.c_float * 22)()
    assert dev.dev_reference_kats(out) == 0
    o = np.array(out[:], dtype=np.float64)
    want = [[1, 0, 0, 0], [0.9238795, 0, 0.3826834, 0], [0.9238795, 0.3826834, 0, 0],
            [0.853553, 0.353553, 0.353553, -0.146447]]
    assert np.abs(o[:16].reshape(4, 4) - np.array(want)).max() < 1e-4   # EXPECT_NEAR 1e-4
    assert o[17] - o[16] <= 1e-5            # Solve4SimplexDuplicatePoint
    assert (np.abs(o[18:21]) < 1e-5).all() and o[21] < 1e-5     # ...AroundOrigin
