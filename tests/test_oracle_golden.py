"""CPU-only: pins the oracle.

* oracle/restate (plain-C restatement) against the known answers in the
  reference's own tests, the published Threefry2x32-20 vectors, and -- where
  oracle/_ref exists -- against the reference's real code (rand::, EntityStore).
* oracle/_ref simulators against the committed golden fixtures.
"""
import ctypes as C
import os

import numpy as np
import pytest

from madrona_amd.simlib import REF_BUILD_DIR, Simulator, ref_lib_path

RESTATE = os.path.join(REF_BUILD_DIR, "liboracle_restate.so")
IDMAP_REF = os.path.join(REF_BUILD_DIR, "libidmap_ref.so")
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def restate(built):
    lib = C.CDLL(RESTATE)
    lib.oracle_split_i.argtypes = [C.c_uint32] * 4 + [C.POINTER(C.c_uint32)] * 2
    lib.oracle_bits32.restype = C.c_uint32
    lib.oracle_bits32.argtypes = [C.c_uint32, C.c_uint32]
    lib.oracle_sample_i32.restype = C.c_int32
    lib.oracle_sample_i32.argtypes = [C.c_uint32, C.c_uint32, C.c_int32, C.c_int32]
    lib.oracle_sample_i32_biased.restype = C.c_int32
    lib.oracle_sample_i32_biased.argtypes = [C.c_uint32, C.c_uint32, C.c_int32, C.c_int32]
    lib.oracle_bits_to_float01.restype = C.c_float
    lib.oracle_bits_to_float01.argtypes = [C.c_uint32]
    lib.oracle_idmap_create.restype = C.c_void_p
    lib.oracle_idmap_create.argtypes = [C.c_uint32, C.c_uint32]
    lib.oracle_idmap_destroy.argtypes = [C.c_void_p]
    lib.oracle_idmap_acquire.restype = C.c_int32
    lib.oracle_idmap_acquire.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.oracle_idmap_release.argtypes = [C.c_void_p, C.c_uint32, C.c_int32]
    lib.oracle_sort_perm.restype = C.c_int32
    lib.oracle_sort_perm.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                     C.c_int32, C.c_void_p, C.c_void_p]
    lib.oracle_gather_column.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int32, C.c_uint32]
    return lib


@pytest.fixture(scope="module")
def refshim(built):
    if not os.path.exists(IDMAP_REF):
        pytest.skip("oracle/_ref not built here (no /root/reference)")
    lib = C.CDLL(IDMAP_REF)
    lib.ref_idmap_create.restype = C.c_void_p
    lib.ref_idmap_create.argtypes = [C.c_uint32]
    lib.ref_idmap_destroy.argtypes = [C.c_void_p]
    lib.ref_idmap_acquire.restype = C.c_int32
    lib.ref_idmap_acquire.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.ref_idmap_release.argtypes = [C.c_void_p, C.c_uint32, C.c_int32, C.c_uint32]
    lib.ref_split_i.argtypes = [C.c_uint32] * 4 + [C.POINTER(C.c_uint32)] * 2
    lib.ref_sample_i32.restype = C.c_int32
    lib.ref_sample_i32.argtypes = [C.c_uint32, C.c_uint32, C.c_int32, C.c_int32]
    lib.ref_sample_i32_biased.restype = C.c_int32
    lib.ref_sample_i32_biased.argtypes = [C.c_uint32, C.c_uint32, C.c_int32, C.c_int32]
    lib.ref_sample_uniform.restype = C.c_float
    lib.ref_sample_uniform.argtypes = [C.c_uint32, C.c_uint32]
    return lib


def split(lib, fn, a, b, i, j=0):
    oa, ob = C.c_uint32(), C.c_uint32()
    getattr(lib, fn)(a, b, i, j, C.byref(oa), C.byref(ob))
    return oa.value, ob.value


# ---- rand ------------------------------------------------------------------
def test_threefry_published_vectors(restate):
    # Threefry2x32-20 known answers (Random123 kat_vectors; also JAX's
    # testThreefry2x32): key, counter -> output
    assert split(restate, "oracle_split_i", 0, 0, 0, 0) == (0x6B200159, 0x99BA4EFE)
    assert split(restate, "oracle_split_i", 0xFFFFFFFF, 0xFFFFFFFF,
                 0xFFFFFFFF, 0xFFFFFFFF) == (0x1CB996FC, 0xBB002BE7)
    assert split(restate, "oracle_split_i", 0x13198A2E, 0x03707344,
                 0x243F6A88, 0x85A308D3) == (0xC4923A9C, 0x483DF7A0)


def test_rand_reference_known_answers(restate):
    # reference tests/rand.cpp:131-141 (RandomRange.UpperLimit)
    assert restate.oracle_bits32(0xFFFFFFFF, 0) == 0xFFFFFFFF
    assert restate.oracle_sample_i32(0xFFFFFFFF, 0, 0, 64) == 63
    assert restate.oracle_sample_i32_biased(0xFFFFFFFF, 0, 0, 64) == 63


def test_rand_reference_range_properties(restate):
    # reference tests/rand.cpp:19-121: ranges, limits found / never exceeded
    init = (5, 0)
    for lo, hi in [(1, 100), (2, 20), (-20, 2), (-30, -10),
                   (-(2 ** 31), -(2 ** 31) + 20)]:
        seen = set()
        for stream in (0, 1):
            key = split(restate, "oracle_split_i", *init, stream)
            for i in range(400):
                k = split(restate, "oracle_split_i", *key, i)
                v = restate.oracle_sample_i32(k[0], k[1], lo, hi)
                assert lo <= v < hi
                seen.add(v)
        assert lo in seen and hi not in seen
    for bits in (0, 1, 0xFFFFFFFF, 0x80000000, 12345678):
        f = restate.oracle_bits_to_float01(bits)
        assert 0.0 <= f < 1.0


def test_rand_restatement_matches_reference(restate, refshim):
    rng = np.random.default_rng(0)
    for _ in range(2000):
        a, b, i, j = (int(x) for x in rng.integers(0, 2 ** 32, 4, dtype=np.uint64))
        assert split(restate, "oracle_split_i", a, b, i, j) == \
            split(refshim, "ref_split_i", a, b, i, j)
        lo = int(rng.integers(-2 ** 31, 2 ** 31 - 2))
        hi = int(rng.integers(lo + 1, min(lo + 2 ** 31, 2 ** 31)))
        assert restate.oracle_sample_i32(a, b, lo, hi) == refshim.ref_sample_i32(a, b, lo, hi)
        assert restate.oracle_sample_i32_biased(a, b, lo, hi) == \
            refshim.ref_sample_i32_biased(a, b, lo, hi)
        assert restate.oracle_bits_to_float01(a ^ b) == refshim.ref_sample_uniform(a, b)


# ---- entity ids ------------------------------------------------------------
def test_idmap_restatement_matches_reference(restate, refshim):
    """Random acquire/release traces over several caches, including overflow
    (>64 frees into one cache -> blocks travel through the global list)."""
    rng = np.random.default_rng(1)
    num_caches = 5
    ours = restate.oracle_idmap_create(num_caches, 1 << 16)
    theirs = refshim.ref_idmap_create(num_caches)
    live = []
    try:
        for step in range(20000):
            phase = (step // 2500) % 2      # grow phases and shrink phases
            do_acquire = rng.random() < (0.7 if phase == 0 else 0.3) or not live
            if do_acquire:
                c = int(rng.integers(0, num_caches))
                g1, g2 = C.c_uint32(), C.c_uint32()
                i1 = restate.oracle_idmap_acquire(ours, c, C.byref(g1))
                i2 = refshim.ref_idmap_acquire(theirs, c, C.byref(g2))
                assert (i1, g1.value) == (i2, g2.value), f"step {step}"
                live.append((i1, g1.value))
            else:
                k = int(rng.integers(0, len(live)))
                eid, gen = live.pop(k)
                # release into a *different* cache now and then (ids migrate)
                c = int(rng.integers(0, num_caches))
                restate.oracle_idmap_release(ours, c, eid)
                refshim.ref_idmap_release(theirs, c, eid, gen)
    finally:
        restate.oracle_idmap_destroy(ours)
        refshim.ref_idmap_destroy(theirs)


def test_idmap_world_block_layout(restate):
    """World w's first entity is id 64*w when every world takes one block in
    world order (what SURVEY.md §7 H2 observed on the reference)."""
    m = restate.oracle_idmap_create(8, 4096)
    try:
        for w in range(8):
            g = C.c_uint32()
            assert restate.oracle_idmap_acquire(m, w, C.byref(g)) == 64 * w
            assert restate.oracle_idmap_acquire(m, w, C.byref(g)) == 64 * w + 1
    finally:
        restate.oracle_idmap_destroy(m)


# ---- sort / compact ----------------------------------------------------------
@pytest.mark.parametrize("n,worlds", [(0, 4), (1, 1), (1000, 7), (5000, 300), (4096, 70000)])
def test_sort_restatement_is_stable_argsort(restate, n, worlds):
    rng = np.random.default_rng(n + worlds)
    keys = rng.integers(0, worlds, n).astype(np.uint32)
    dead = rng.random(n) < 0.2
    keys[dead] = 0xFFFFFFFF
    perm = np.empty(max(n, 1), np.int32)
    offs = np.empty(worlds, np.int32)
    cnts = np.empty(worlds, np.int32)
    n_out = restate.oracle_sort_perm(keys.ctypes.data, n, 1, perm.ctypes.data, worlds,
                                     offs.ctypes.data, cnts.ctypes.data)
    live = np.nonzero(~dead)[0]
    expect = live[np.argsort(keys[live], kind="stable")]
    assert n_out == len(expect)
    assert np.array_equal(perm[:n_out], expect)
    assert np.array_equal(cnts, np.bincount(keys[live], minlength=worlds)[:worlds])
    starts = np.concatenate([[0], np.cumsum(cnts)[:-1]])
    nonempty = cnts > 0
    assert np.array_equal(offs[nonempty], starts[nonempty])
    assert (offs[~nonempty] == n_out).all()

    payload = rng.integers(0, 255, (n, 20), dtype=np.uint8)
    out = np.empty((max(n_out, 1), 20), np.uint8)
    restate.oracle_gather_column(payload.ctypes.data, out.ctypes.data, perm.ctypes.data,
                                 n_out, 20)
    assert np.array_equal(out[:n_out], payload[expect])


# ---- reference simulators vs committed golden fixtures -------------------------
def _replay_against_golden(path, lib_path):
    from golden.make_golden import CASES, actions_for
    name = os.path.basename(path)[:-4]
    sim, worlds, seed, flags, checkpoints = CASES[name]
    gold = np.load(path)
    with Simulator(lib_path, worlds, seed=seed, num_workers=1, flags=flags) as s:
        for step in range(1, max(checkpoints) + 1):
            actions = actions_for(sim, step, worlds)
            if actions is not None:
                s.write_tensor("action", actions)
            s.step(1)
            if step in checkpoints:
                for col, (rows, counts) in s.dump_all().items():
                    assert np.array_equal(counts, gold[f"s{step}/{col}/counts"]), (step, col)
                    assert np.array_equal(rows, gold[f"s{step}/{col}/rows"]), (step, col)


@pytest.mark.parametrize("name", ["cartpole_w64", "escape_room_w16",
                                  "sort_stress_w33", "escape_room_phys_w8",
                                  "hideseek_w8", "ball_pit_w8"])
def test_reference_backend_reproduces_golden(built, name):
    from golden.make_golden import CASES
    sim = CASES[name][0]
    if not os.path.exists(ref_lib_path(sim)):
        pytest.skip("oracle/_ref not built here (no /root/reference)")
    _replay_against_golden(os.path.join(GOLDEN, f"{name}.npz"), ref_lib_path(sim))
