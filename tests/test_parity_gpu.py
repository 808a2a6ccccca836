"""GPU parity tests (``pytest -m gpu`` on the MI355X box).  Everything goes
through the C ABI (simulator .so -> MWCudaExecutor shim -> libmadrona_hip.so).

Oracles, in order of preference:
  1. the reference CPU backend itself (oracle/_ref/*.so travels to the GPU box)
     stepped in lock step, every dumped column compared bit for bit;
  2. the committed golden fixtures generated from it (tests/golden/*.npz);
  3. size-independent properties at BASELINE's full sizes (sortedness, id
     uniqueness, entity-store consistency, partition invariance).
Floating-point columns are compared BIT-EXACT too: the simulators only use
+ - * / sqrt and both sides are built with -ffp-contract=off (tighter than the
1e-5 relative tolerance BASELINE.json allows).
"""
import os

import numpy as np
import pytest

from madrona_amd.simlib import Simulator, hip_lib_path, ref_lib_path
from parity_utils import compare_columns, run_pair

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _need_ref(sim):
    if not os.path.exists(ref_lib_path(sim)):
        pytest.skip("oracle/_ref missing on this box")


def _escape_actions(seed, grab=False, agents=2):
    rng = np.random.default_rng(seed)

    def feed(ref, hip, step):
        W = ref.num_worlds
        shape = (W, agents)
        a = np.stack([rng.integers(0, 4, shape), rng.integers(0, 8, shape),
                      rng.integers(-2, 3, shape),
                      rng.integers(0, 2, shape) if grab else
                      np.zeros(shape, int)],
                     -1).astype(np.int32)
        ref.write_tensor("action", a)
        hip.write_tensor("action", a)
    return feed


# ---- 1. lock-step against the reference CPU backend ---------------------------
@pytest.mark.parametrize("worlds", [1, 64, 1000])
def test_cartpole_lockstep(built, worlds):
    _need_ref("cartpole")
    probs, step = run_pair("cartpole", worlds, 250, check_every=10)
    assert not probs, (step, probs[:3])


@pytest.mark.parametrize("worlds,denom,steps", [(1, 5, 120), (64, 50, 300),
                                                (300, 200, 250), (2048, 100, 60)])
def test_escape_room_lockstep(built, worlds, denom, steps):
    """Resets (destroy 27 / create 27 entities per reset), compaction of three
    archetypes every step, entity ids and generations, 30 columns."""
    _need_ref("escape_room")
    probs, step = run_pair("escape_room", worlds, steps, flags=denom,
                           check_every=1 if worlds <= 64 else 10,
                           actions=_escape_actions(worlds), check_init=False)
    assert not probs, (step, probs[:3])


def test_escape_room_external_reset_and_timeouts(built):
    """No random resets: episodes end by the 200-step timeout and by the
    exported reset tensor being written from outside (as a trainer would)."""
    _need_ref("escape_room")
    W = 32
    feed = _escape_actions(3)

    def actions(ref, hip, step):
        feed(ref, hip, step)
        if step in (7, 90):
            r = np.zeros((W, 1), np.int32)
            r[::3] = 1
            ref.write_tensor("reset", r)
            hip.write_tensor("reset", r)

    probs, step = run_pair("escape_room", W, 230, flags=0, actions=actions,
                           check_init=False, check_every=5)
    assert not probs, (step, probs[:3])


@pytest.mark.parametrize("worlds,denom,steps", [(1, 0, 230), (16, 40, 150),
                                                (256, 100, 80), (1024, 150, 30)])
def test_escape_room_physics_lockstep(built, worlds, denom, steps):
    """BASELINE config 3 shape: BVH build/refit, broadphase candidates,
    SAT narrowphase (hull-hull, hull-plane), XPBD contacts + fixed joints
    (grab action), resets re-registering every body.  fp32 state is compared
    bit for bit (stricter than the 1e-5 relative bound): contact order is the
    CPU backend's, so the Gauss-Seidel solve sees the same sequence."""
    _need_ref("escape_room_phys")
    probs, step = run_pair("escape_room_phys", worlds, steps, flags=denom,
                           check_every=1 if worlds <= 16 else 10,
                           actions=_escape_actions(worlds + 1, grab=True),
                           check_init=False)
    assert not probs, (step, probs[:3])


@pytest.mark.parametrize("max_bodies", [64, 128, 1000])
def test_escape_room_physics_kernel_variants(built, monkeypatch, max_bodies):
    """The other instantiations of the fused step: LDS blocks sized for 64 / 128
    bodies per world, and (> 128) the variant that works out of HBM."""
    _need_ref("escape_room_phys")
    monkeypatch.setenv("MADRONA_MWHIP_PHYS_MAX_BODIES", str(max_bodies))
    probs, step = run_pair("escape_room_phys", 48, 60, flags=25,
                           check_every=5, actions=_escape_actions(7, grab=True),
                           check_init=False)
    assert not probs, (step, probs[:3])


@pytest.mark.parametrize("worlds,crowded,block", [(256, 3, ""), (64, 31, ""),
                                                  (48, 7, "64"), (48, 8, "128")])
def test_one_crowded_world_falls_back_to_the_hbm_step(built, monkeypatch, worlds, crowded,
                                                      block):
    """a world the LDS instantiation cannot hold is stepped out of
    HBM in the same step instead of raising kErrPhysics (the reference has no
    cap on a world's bodies: broadphase.cpp:892-1052).  ball_pit with ONE world
    of 168 rigid bodies among worlds of 18 (flags bit 26 + the crowd size): the
    step node is sized for the bulk of the worlds -- the 32-body block, two
    worlds per wavefront -- and the crowded world, whose partner in the
    wavefront stays in the LDS kernel, goes through physicsStepKernel in
    fallback mode.  Lock step with the reference, no error flag (a raised flag
    fails the step that raised it).  `block`: the same under the 64- and 128-body
    blocks (one world per wavefront, MADRONA_MWHIP_PHYS_MAX_BODIES), which the
    crowded world outgrows as well."""
    _need_ref("ball_pit")
    if block:
        monkeypatch.setenv("MADRONA_MWHIP_PHYS_MAX_BODIES", block)
    monkeypatch.setenv("MADRONA_MWHIP_MAX_CANDIDATES_PER_WORLD", "4096")
    monkeypatch.setenv("MADRONA_MWHIP_MAX_CONTACTS_PER_WORLD", "1024")
    flags = (1 << 26) | (crowded << 27) | (150 << 16)
    probs, step = run_pair("ball_pit", worlds, 50, flags=flags, check_every=5,
                           check_init=False, ref_workers=0)
    assert not probs, (step, probs[:3])
    with Simulator(hip_lib_path("ball_pit"), worlds, flags=flags) as s:
        s.step(20)
        names = [k["name"] for k in s.profile(2)]
        assert "physics:worldStep(LDS)" in names, names
        assert "physics:worldStep(fallback)" in names, names
        counts = s.dump_all()["MovableObject.Position"][1]
        assert counts[crowded] == 14 + 150 and counts.sum() == 14 * worlds + 150


def test_a_world_with_more_movable_bodies_than_solver_slots_falls_back(built):
    """Round 6: the two-worlds-per-wavefront block keeps solver records for 20 of
    a world's 32 bodies (an inert static body's records are its pose).  ball_pit
    with ONE world of 14 + 12 movable bodies (30 rigid bodies: it fits the
    32-body block) among worlds of 14: the world is listed and stepped by the
    HBM kernel in the same step, its partner in the wavefront stays.  Lock step
    with the reference."""
    _need_ref("ball_pit")
    worlds, crowded, extra = 128, 5, 12
    flags = (1 << 26) | (crowded << 27) | (extra << 16)
    probs, step = run_pair("ball_pit", worlds, 60, flags=flags, check_every=5,
                           check_init=False, ref_workers=0)
    assert not probs, (step, probs[:3])
    with Simulator(hip_lib_path("ball_pit"), worlds, flags=flags) as s:
        s.step(10)
        counts = s.dump_all()["MovableObject.Position"][1]
        assert counts[crowded] == 14 + extra and counts.max() == 14 + extra
        stats = {k["name"]: k for k in s.profile(2)}
        assert "physics:worldStep(LDS)" in stats and \
            "physics:worldStep(fallback)" in stats, list(stats)
        # (an empty fallback list costs the launch its floor, ~4 us: this one
        # steps a 30-body world out of HBM)
        assert stats["physics:worldStep(fallback)"]["avg_us"] > 20.0, \
            stats["physics:worldStep(fallback)"]


@pytest.mark.parametrize("cap,lanes", [("20", ""), ("1", ""), ("20", "64")])
def test_contacts_beyond_the_lds_block_fall_back(built, monkeypatch, cap, lanes):
    """The other capacity of the LDS block: more contacts in a substep than it
    has room for (64 with the 32-body block).  19-body ball_pit worlds forced
    into a tight pit pile up more; seed 11 is the one that used to raise
    kErrTableOverflow before the constant was raised.  With the
    capacity cut to 20 (MADRONA_MWHIP_PHYS_LDS_CONTACTS) the pile overflows
    within a few steps in many worlds -- they take the HBM step for that step
    and stay in lock step; with it at 1 every world with a contact does."""
    _need_ref("ball_pit")
    monkeypatch.setenv("MADRONA_MWHIP_PHYS_LDS_CONTACTS", cap)
    if lanes:
        monkeypatch.setenv("MADRONA_MWHIP_PHYS_LANES", lanes)
    probs, step = run_pair("ball_pit", 96, 60, flags=40, seed=11, check_every=5,
                           check_init=False)
    assert not probs, (step, probs[:3])


@pytest.mark.parametrize("worlds,denom,steps", [(1, 0, 260), (16, 40, 150),
                                                (64, 0, 300), (256, 100, 60),
                                                (1024, 200, 30), (4096, 150, 20)])
def test_hideseek_lockstep(built, worlds, denom, steps):
    """BASELINE config 4 shape: 29 rigid bodies per world in three archetypes
    (5 agents, 9 boxes, 2 wedge-shaped ramps, 12 walls, plane) -> the 64-body
    LDS instantiation of the fused step; hull-hull SAT between different hulls
    (cube / wedge: triangle faces, non axis-aligned edge pairs), bodies that
    switch between Dynamic and Static (lock action), line-of-sight rays and a
    lidar through BVH::traceRay, resets that rebuild the BVH."""
    _need_ref("hideseek")
    probs, step = run_pair("hideseek", worlds, steps, flags=denom,
                           check_every=1 if worlds <= 16 else 10,
                           actions=_escape_actions(worlds + 3, grab=True,
                                                   agents=5),
                           check_init=False)
    assert not probs, (step, probs[:3])


@pytest.mark.parametrize("max_bodies", [32, 128, 1000])
def test_hideseek_kernel_variants(built, monkeypatch, max_bodies):
    """29 bodies also fit the 32-body block exactly; 128 and the HBM variant."""
    _need_ref("hideseek")
    monkeypatch.setenv("MADRONA_MWHIP_PHYS_MAX_BODIES", str(max_bodies))
    probs, step = run_pair("hideseek", 48, 60, flags=25, check_every=5,
                           actions=_escape_actions(9, grab=True, agents=5),
                           check_init=False)
    assert not probs, (step, probs[:3])


@pytest.mark.parametrize("worlds,denom,steps", [(1, 0, 320), (16, 30, 200),
                                                (256, 60, 100), (2048, 150, 40)])
def test_ball_pit_lockstep(built, worlds, denom, steps):
    """Sphere primitives: sphere-sphere, sphere-plane and sphere-hull (GJK with
    the signed-volume sub-solvers, then the SAT fallback for centres inside the
    hull) on the device, next to box / wedge hulls and an object made of two
    hull primitives (one candidate per primitive pair; its hulls do not fit the
    LDS arena next to the others and are read from HBM), kicked around by
    random forces -- a chaotic pile, so an ulp anywhere shows within a few
    steps."""
    _need_ref("ball_pit")
    probs, step = run_pair("ball_pit", worlds, steps, flags=denom,
                           check_every=1 if worlds <= 16 else 10,
                           check_init=False)
    assert not probs, (step, probs[:3])


@pytest.mark.parametrize("max_bodies", [64, 128, 1000])
def test_ball_pit_kernel_variants(built, monkeypatch, max_bodies):
    """Spheres and the two-primitive object through the other instantiations of
    the fused step (the > 128 variant reads primitives and hulls from HBM)."""
    _need_ref("ball_pit")
    monkeypatch.setenv("MADRONA_MWHIP_PHYS_MAX_BODIES", str(max_bodies))
    probs, step = run_pair("ball_pit", 64, 80, flags=30, check_every=5,
                           check_init=False)
    assert not probs, (step, probs[:3])


@pytest.mark.parametrize("sim,worlds,steps,denom,agents", [
    ("escape_room_phys", 49, 90, 25, 2),       # odd: the last wavefront has one world
    ("escape_room_phys", 1, 60, 10, 2),
    ("escape_room_phys", 1024, 40, 100, 2),
    ("ball_pit", 33, 120, 40, 0),              # spheres, two-primitive object, the
                                               # coins' faces that outgrow the LDS
                                               # scratch (HBM hull scratch, one
                                               # lane at a time), 8 joints
    ("ball_pit", 256, 60, 30, 0)])
@pytest.mark.parametrize("lanes", ["32", "64"])
def test_physics_lanes_per_world(built, monkeypatch, sim, worlds, steps, denom, agents,
                                 lanes):
    """physicsStepLdsKernel<32, 32> (the default for worlds of at most 32 bodies):
    two worlds per wavefront, one per half -- every wave-level primitive on groups
    of 32 lanes, the halves diverging wherever their worlds differ (candidate /
    contact counts, hull-hull pairs, solver levels) -- and <32, 64>, one world per
    wavefront.  Same bit-for-bit bar for both."""
    _need_ref(sim)
    monkeypatch.setenv("MADRONA_MWHIP_PHYS_LANES", lanes)
    probs, step = run_pair(sim, worlds, steps, flags=denom,
                           check_every=1 if worlds <= 64 else 10,
                           actions=_escape_actions(worlds + 5, grab=True, agents=agents)
                           if agents else None,
                           check_init=False, ref_workers=0 if worlds > 64 else 1)
    assert not probs, (step, probs[:3])


@pytest.mark.parametrize("extra,kernel", [(60, "LDS<128>"), (140, "HBM")])
def test_ball_pit_crowd(built, monkeypatch, extra, kernel):
    """Worlds of 79 and 159 rigid bodies (ball_pit's crowd mode): more bodies
    than lanes, so the body / candidate / contact loops of the step kernels run
    several 64-wide chunks -- the 128-body LDS instantiation and the variant
    that works out of HBM -- and the BVH (> 64 leaves) is rebuilt in place
    instead of in the LDS staging of bvhUpdateKernel.  (Their BVH arrays also
    outgrow the default persistent region, and their constructors create more
    rows than the default 64 per world: the executor sizes the region from what
    the first constructor pass asked for, and grows the full tables.)"""
    _need_ref("ball_pit")
    monkeypatch.setenv("MADRONA_MWHIP_MAX_CANDIDATES_PER_WORLD", "2048")
    monkeypatch.setenv("MADRONA_MWHIP_MAX_CONTACTS_PER_WORLD", "1024")
    # (resets only for the smaller crowd: a world that frees more than two
    # blocks of entity ids at once spills them into the CPU backend's global
    # free list, the id-renaming regime of DESIGN.md section 5 -- and body ids
    # order the candidate pairs)
    denom = 25 if extra <= 60 else 0
    probs, step = run_pair("ball_pit", 24, 70, flags=(extra << 16) | denom,
                           check_every=5, check_init=False)
    assert not probs, (kernel, step, probs[:3])


def _candidate_pairs(dump, arch_names):
    """CandidateCollision rows -> (world, entity id a, entity id b, aPrim, bPrim).
    A Loc's row is world-local on the CPU backend and global on the GPU; both are
    resolved through the dumped Entity columns (grouped by world, world order)."""
    cand, cand_counts = dump["Candidates.CandidateCollision"]
    cand = cand.view(np.int32).reshape(-1, 6)      # a.arch a.row b.arch b.row aPrim bPrim
    tables = {}
    for arch_id, name in arch_names.items():
        ents, counts = dump[f"{name}.Entity"]
        ids = ents.view(np.int32).reshape(-1, 2)[:, 1]
        tables[arch_id] = (ids, np.concatenate([[0], np.cumsum(counts)]))
    out = []
    world_of_row = np.repeat(np.arange(len(cand_counts)), cand_counts)
    for row, w in zip(cand, world_of_row):
        pair = []
        for arch, r in ((row[0], row[1]), (row[2], row[3])):
            ids, starts = tables[int(arch)]
            local_guess = starts[w] + r           # CPU: world-local row
            global_guess = r                      # GPU: global row
            pair.append((int(ids[local_guess]) if local_guess < len(ids) else -1,
                         int(ids[global_guess]) if global_guess < len(ids) else -1))
        out.append((int(w), pair, int(row[4]), int(row[5])))
    return out, cand_counts


@pytest.mark.parametrize("worlds", [1, 7, 200])
def test_standalone_broadphase_candidates(built, worlds):
    """setupStandaloneBroadphaseOverlapTasks: the table-based candidate path
    (count -> exclusive-scan node -> fill into the CandidateTemporary archetype,
    sorted by world).  Same pairs, in the same order, as the CPU backend's
    per-world traversal; BVH rebuilt every 16 steps."""
    _need_ref("broadphase_only")
    with Simulator(ref_lib_path("broadphase_only"), worlds, seed=9, num_workers=1) as ref, \
            Simulator(hip_lib_path("broadphase_only"), worlds, seed=9) as hip:
        for step in range(1, 41):
            ref.step(1)
            hip.step(1)
            rd, hd = ref.dump_all(), hip.dump_all()
            for col in ("Box.Entity", "Box.Position", "Box.LeafID", "Pillar.Entity"):
                assert np.array_equal(rd[col][0], hd[col][0]), (step, col)

            # archetype ids: taken from the first candidate rows of the reference
            # (Box rows are created after Pillar rows; ids are registration order)
            rc = rd["Candidates.CandidateCollision"][0].view(np.int32).reshape(-1, 6)
            arch_ids = sorted(set(rc[:, 0]) | set(rc[:, 2]))
            assert len(arch_ids) == 2
            names = {arch_ids[0]: "Box", arch_ids[1]: "Pillar"}

            ref_pairs, ref_counts = _candidate_pairs(rd, names)
            hip_pairs, hip_counts = _candidate_pairs(hd, names)
            assert np.array_equal(ref_counts, hip_counts), step
            assert ref_counts.sum() > 0
            for (rw, rp, ra, rb), (hw, hp, ha, hb) in zip(ref_pairs, hip_pairs):
                assert rw == hw and ra == ha and rb == hb
                # CPU rows are world-local, GPU rows global
                assert rp[0][0] == hp[0][1] and rp[1][0] == hp[1][1], (step, rw)


@pytest.mark.parametrize("worlds", [5, 97])
def test_rays_sharing_an_origin_big_and_uneven_trees(built, worlds):
    """BVH::traceRayShared (32 lanes per sensor, the ray-independent half of a
    leaf test once per leaf in LDS) against the reference's BVH::traceRay, ray by
    ray: distance, entity id and normal bit for bit.  broadphase_only in ray mode:
    10 .. 100 boxes per world -- trees of more than 64 leaves (two windows of
    the shared scratch) next to small ones -- and 1 .. 3 sensors per world, so
    that the two halves of a wavefront hold sensors of different worlds: different
    trees, different leaf counts, different window counts.  The BVH is rebuilt
    every 16 steps (the pending-rebuild walk in between is exercised too)."""
    _need_ref("broadphase_only")
    with Simulator(ref_lib_path("broadphase_only"), worlds, seed=3, num_workers=1,
                   flags=1) as ref, \
            Simulator(hip_lib_path("broadphase_only"), worlds, seed=3, flags=1) as hip:
        hits = 0
        for step in range(1, 36):
            ref.step(1)
            hip.step(1)
            # (crowded worlds hold thousands of candidate pairs)
            rd, hd = ref.dump_all(8192), hip.dump_all(8192)
            for col in ("Box.Position", "Box.LeafID", "Sensor.Position"):
                assert np.array_equal(rd[col][0], hd[col][0]), (step, col)
            assert np.array_equal(rd["Sensor.RayFan"][1], hd["Sensor.RayFan"][1])
            rf = rd["Sensor.RayFan"][0].view(np.int32).reshape(-1, 160)
            hf = hd["Sensor.RayFan"][0].view(np.int32).reshape(-1, 160)
            bad = np.argwhere(rf != hf)
            assert bad.size == 0, (step, bad[:5], rf[tuple(bad[0])], hf[tuple(bad[0])])
            hits += int((rf[:, 32:64] >= 0).sum())
        assert rd["Box.LeafID"][1].max() > 64 and rd["Box.LeafID"][1].min() < 32
        assert hits > 35 * worlds * 8       # the rays do meet boxes


@pytest.mark.parametrize("sim,worlds,steps,flags", [
    ("escape_room_phys", 200, 80, 10),     # 28 leaves, a world in ten resets per step
    ("hideseek", 150, 80, 8),              # 29 leaves, wedges
    ("ball_pit", 256, 80, 12),             # 18 leaves
    ("ball_pit", 40, 40, (40 << 16) | 6),  # 58 leaves: close to the staging's 64
    ("broadphase_only", 97, 40, 1),        # 14 .. 104 leaves, rebuilt every 16 steps
                                           # (> 64: the in-place build)
])
def test_breadth_first_bvh_rebuild_equals_the_stack_machine(built, monkeypatch, sim,
                                                            worlds, steps, flags):
    """BVH::rebuildStagedSegmented (every range of a level split at once; node
    ids, bounds and traversal order derived from the range records) against
    BVH::rebuildStagedWave (the reference's stack machine on a wavefront) on the
    same leaves, word for word: nodes, leaf parents, sorted order, traversal
    order (bvhUpdateKernel<true>, MADRONA_MWHIP_BVH_CHECK=1; a difference
    raises kErrPhysics and fails the step) -- in lock step with the reference
    on top.  Frequent resets: every reset is a rebuild."""
    _need_ref(sim)
    monkeypatch.setenv("MADRONA_MWHIP_BVH_REFRESH", "1")
    monkeypatch.setenv("MADRONA_MWHIP_BVH_CHECK", "1")
    monkeypatch.setenv("MADRONA_MWHIP_MAX_CANDIDATES_PER_WORLD", "2048")
    monkeypatch.setenv("MADRONA_MWHIP_MAX_CONTACTS_PER_WORLD", "1024")
    agents = {"escape_room_phys": 2, "hideseek": 5}.get(sim, 0)
    if sim == "broadphase_only":
        with Simulator(ref_lib_path(sim), worlds, seed=3, num_workers=1,
                       flags=flags) as ref, \
                Simulator(hip_lib_path(sim), worlds, seed=3, flags=flags) as hip:
            for step in range(1, steps + 1):
                ref.step(1)
                hip.step(1)
                rd, hd = ref.dump_all(8192), hip.dump_all(8192)
                for col in ("Box.Position", "Sensor.RayFan"):
                    assert np.array_equal(rd[col][0], hd[col][0]), (step, col)
        return
    probs, step = run_pair(sim, worlds, steps, flags=flags, check_every=10,
                           actions=_escape_actions(worlds + 2, grab=True, agents=agents)
                           if agents else None,
                           check_init=False, ref_workers=0)
    assert not probs, (step, probs[:3])


@pytest.mark.parametrize("sim,worlds,steps,flags", [("escape_room_phys", 96, 50, 25),
                                                    ("sort_stress", 200, 30, 0)])
def test_eager_replay_is_the_same_step(built, monkeypatch, sim, worlds, steps, flags):
    """MADRONA_MWHIP_EAGER=1 (the launches of a step issued one by one
    on the stream instead of the instantiated hipGraph -- a measurement switch):
    the same kernels in the same order, so the same results."""
    _need_ref(sim)
    monkeypatch.setenv("MADRONA_MWHIP_EAGER", "1")
    probs, step = run_pair(sim, worlds, steps, flags=flags, check_every=10,
                           actions=_escape_actions(worlds + 4, grab=True)
                           if sim == "escape_room_phys" else None,
                           check_init=False)
    assert not probs, (step, probs[:3])


@pytest.mark.parametrize("env", [{"MADRONA_MWHIP_GROUP": "0"},
                                 {"MADRONA_MWHIP_GROUP_MAX_VGPRS": "1000"}])
@pytest.mark.parametrize("sim,hip_sim,worlds,steps,agents", [
    ("escape_room_phys", "escape_room_phys_portable", 96, 40, 2),
    ("hideseek", "hideseek", 64, 40, 5)])
def test_shared_launches_are_the_same_step(built, monkeypatch, env, sim, hip_sim, worlds,
                                           steps, agents):
    """ParallelFor nodes that named the same dependencies share one
    launch (every other lock step runs that way).  The two switches around it:
    every node its own launch, and no register limit on the members -- the
    portable lidar system (a BVH traversal per row, 121 VGPRs) then runs through
    the shared kernel's function pointer next to the observations."""
    _need_ref(sim)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    probs, step = run_pair(sim, worlds, steps, check_every=10,
                           actions=_escape_actions(worlds + 6, grab=True, agents=agents),
                           check_init=False, ref_workers=0, hip_sim=hip_sim)
    assert not probs, (step, probs[:3])


def test_ball_pit_hinge_joints(built):
    """Hinge joints (reference src/physics/xpbd.cpp:686-693 and the two
    orientation constraints before it).  A hinged chain is not stable on the
    reference itself (its positional correction pushes the anchors apart:
    positions grow ~5x per step and the CPU backend stops making progress after
    a dozen steps), so the pin is the first steps of that trajectory, bit for
    bit, 20 substeps of the joint solve with contacts around it."""
    _need_ref("ball_pit")
    probs, step = run_pair("ball_pit", 96, 5, flags=1 << 24)
    assert not probs, (step, probs[:3])


def test_ball_pit_entity_aabb_overlap(built):
    """PhysicsSystem::checkEntityAABBOverlap (reference src/physics/
    physics.cpp:159-) decides the direction of every kick."""
    _need_ref("ball_pit")
    probs, step = run_pair("ball_pit", 128, 150, flags=(1 << 25) | 60,
                           check_every=10)
    assert not probs, (step, probs[:3])
    # the test really steered something: the trajectory differs from the plain one
    with Simulator(hip_lib_path("ball_pit"), 16, flags=1 << 25) as a, \
            Simulator(hip_lib_path("ball_pit"), 16, flags=0) as b:
        a.step(100)
        b.step(100)
        pa = a.dump_all()["MovableObject.Position"][0]
        pb = b.dump_all()["MovableObject.Position"][0]
        assert not np.array_equal(pa, pb)


@pytest.mark.parametrize("worlds", [1, 2, 33, 255, 256, 300, 5000])
def test_sort_stress_lockstep(built, worlds):
    """Ragged / empty worlds, every gather width (1..240 B columns),
    temporaries + ClearTmp, 1- and 2-pass world sorts (255 vs 256 worlds)."""
    _need_ref("sort_stress")
    steps = 60 if worlds <= 300 else 25
    probs, step = run_pair("sort_stress", worlds, steps, seed=7,
                           check_every=1 if worlds <= 64 else 5)
    assert not probs, (step, probs[:3])


# (60 steps: the reference's angular update feeds the WORLD-space angular velocity
# into the body-space gyroscopic term, tgs.cpp:135-136 -- reproduced -- and a
# tumbling slab gains energy until its rotation is NaN after ~70 steps, on both
# backends; x86 and gfx950 then disagree about the NaN's sign bit)
@pytest.mark.parametrize("worlds,steps", [(1, 60), (33, 60), (1000, 50)])
def test_tgs_solver_lockstep(built, worlds, steps):
    """PhysicsSystem with Solver::TGS (reference src/physics/tgs.cpp:
    integrateVelocities :93-145 with the body-space gyroscopic term,
    integratePositions :172-196, task wiring :225-304): dynamic, kinematic and
    static bodies, a cube and an anisotropic slab, random forces and torques
    every step -- positions, rotations and velocities bit for bit against the
    reference built with the same solver switch."""
    _need_ref("tgs_drop")
    probs, step = run_pair("tgs_drop", worlds, steps, seed=9,
                           check_every=1 if worlds <= 64 else 10,
                           ref_workers=0 if worlds > 64 else 1)
    assert not probs, (step, probs[:3])


# Every way through the sort node, on the same workloads:
#   compact 0          radix chain only (histogram + key passes + gather)
#   compact 1          default: compaction chain where nothing but world sorts
#                      reorders the table, radix chain elsewhere
#   compact 2          compaction chain for EVERY world sort: also tables that a
#                      key sort or ClearTmp leaves with no sorted prefix (the
#                      whole table is then "tail": one workgroup sorts it)
#   grid N             key-pass / scatter grids capped at N workgroups: tiles
#                      beyond the grid are taken in further rounds
#                      (sort_archetype.hip: tile = workgroup + round * grid)
#   wide 0             gather word by word instead of in 16-byte chunks of the
#                      destination
# (compaction mode, grid cap, wide gather, trailing misc ops ride on the chain)
SORT_CHAINS = [("0", "", "1", "1"), ("2", "", "1", "1"), ("2", "3", "1", "1"),
               ("0", "2", "1", "1"), ("1", "1", "1", "1"), ("1", "", "0", "1"),
               ("0", "", "0", "1"), ("2", "", "1", "0"), ("0", "", "1", "0")]


@pytest.mark.parametrize("compact,grid,wide,carry", SORT_CHAINS)
@pytest.mark.parametrize("sim,worlds,steps,denom", [
    ("sort_stress", 1500, 25, 0), ("sort_stress", 37, 40, 0),
    ("escape_room", 2048, 60, 30)])
def test_sort_chains_lockstep(built, monkeypatch, sim, worlds, steps, denom, compact,
                              grid, wide, carry):
    _need_ref(sim)
    monkeypatch.setenv("MADRONA_MWHIP_SORT_COMPACT", compact)
    monkeypatch.setenv("MADRONA_MWHIP_SORT_CARRIES_MISC", carry)
    monkeypatch.setenv("MADRONA_MWHIP_GATHER_WIDE", wide)
    if grid:
        monkeypatch.setenv("MADRONA_MWHIP_SORT_MAX_GRID", grid)
    if worlds < 100:
        # small tables: force the chains instead of the single-launch sort
        monkeypatch.setenv("MADRONA_MWHIP_SORT_SMALL", "0")
    probs, step = run_pair(sim, worlds, steps, seed=7, flags=denom, check_every=5,
                           actions=_escape_actions(9) if sim == "escape_room" else None,
                           check_init=False, ref_workers=0)
    assert not probs, (step, probs[:3])


@pytest.mark.parametrize("compact", ["1", "2"])
def test_compaction_chain_heavy_churn(built, monkeypatch, compact):
    """The compaction chain where it is the default: the Escape Room with a reset
    every ~3 steps per world (a third of every table destroyed and re-created
    per step: long tails, many tail rows landing in one prefix tile, worlds whose
    whole prefix range died), 4096 worlds, against the reference.  In mode 1
    the executor sends such a table back to the radix chain after three long
    tails (graphs rebuilt mid-run); mode 2 keeps the compaction chain on it."""
    _need_ref("escape_room")
    monkeypatch.setenv("MADRONA_MWHIP_SORT_COMPACT", compact)
    probs, step = run_pair("escape_room", 4096, 40, seed=21, flags=3, check_every=4,
                           actions=_escape_actions(10), check_init=False,
                           ref_workers=0)
    assert not probs, (step, probs[:3])


def test_compaction_tile_owning_more_rows_than_it_orders_in_lds(built):
    """the by-landing-points path of the compaction chain when ONE
    scatter tile owns more tail rows than its LDS buffer (kOwnCap = 2048) -- the
    rank-by-counting fallback -- and the 16-run countdown (landsBlocked) that
    sends the following runs down the sorted-tail path before the landing-point
    path is taken again.  sort_stress in burst mode: 30 000 worlds of one item,
    the first hundred create 30 items each in ONE step (3 000 new rows behind a
    30 000-row sorted prefix, all landing in the first tile), twice, 27 steps
    apart, with a few worlds churning in between so that every step sorts."""
    _need_ref("sort_stress")
    probs, step = run_pair("sort_stress", 30000, 45, seed=5, flags=8,
                           check_every=1, check_init=False, ref_workers=0)
    assert not probs, (step, probs[:3])


def test_sort_stress_cold_start_runtime_id_blocks(built):
    """Worlds that start empty take their first block of entity ids at run
    time.  The CPU backend numbers such blocks in world-major order within a
    step, the GPU from a static per-world partition: ids differ
    by a renaming, every other column must still match bit for bit, and the
    HIP run must be reproducible."""
    _need_ref("sort_stress")
    W, steps = 255, 30
    dumps = []
    for _ in range(2):
        with Simulator(hip_lib_path("sort_stress"), W, seed=7, flags=1) as s:
            s.step(steps)
            dumps.append(s.dump_all())
    assert not compare_columns(dumps[0], dumps[1])

    with Simulator(ref_lib_path("sort_stress"), W, seed=7, num_workers=1, flags=1) as r:
        r.step(steps)
        ref = r.dump_all()
    hip = dumps[0]
    ref_ids = ref.pop("Item.Entity")[0].view(np.int32).reshape(-1, 2)
    hip_ids = hip.pop("Item.Entity")[0].view(np.int32).reshape(-1, 2)
    assert not compare_columns(ref, hip)
    # same generations, ids related by a bijection
    assert np.array_equal(ref_ids[:, 0], hip_ids[:, 0])
    pairs = np.unique(np.stack([ref_ids[:, 1], hip_ids[:, 1]], 1), axis=0)
    assert len(np.unique(pairs[:, 0])) == len(pairs) == len(np.unique(pairs[:, 1]))
    assert (ref_ids[:, 1] != hip_ids[:, 1]).any()  # the regime was actually hit


def test_sort_three_pass_world_ids(built):
    """> 65534 worlds needs a third radix pass (reference sort_archetype.cpp:
    1432-1438)."""
    _need_ref("sort_stress")
    probs, step = run_pair("sort_stress", 66000, 6, seed=11, check_every=3)
    assert not probs, (step, probs[:3])


# ---- 2. committed golden fixtures ------------------------------------------------
@pytest.mark.parametrize("name", ["cartpole_w64", "escape_room_w16",
                                  "sort_stress_w33", "escape_room_phys_w8",
                                  "hideseek_w8", "ball_pit_w8"])
def test_hip_matches_golden(built, name):
    from golden.make_golden import CASES, actions_for
    sim, worlds, seed, flags, checkpoints = CASES[name]
    gold = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    with Simulator(hip_lib_path(sim), worlds, seed=seed, flags=flags) as s:
        for step in range(1, max(checkpoints) + 1):
            actions = actions_for(sim, step, worlds)
            if actions is not None:
                s.write_tensor("action", actions)
            s.step(1)
            if step in checkpoints:
                checked = 0
                for col, (rows, counts) in s.dump_all().items():
                    if f"s{step}/{col}/rows" not in gold.files:
                        # (a column only the HIP backend's global tables have)
                        assert col.endswith(".WorldID"), col
                        continue
                    assert np.array_equal(counts, gold[f"s{step}/{col}/counts"]), (step, col)
                    assert np.array_equal(rows, gold[f"s{step}/{col}/rows"]), (step, col)
                    checked += 1
                assert checked == sum(k.startswith(f"s{step}/") and k.endswith("/rows")
                                      for k in gold.files)


# ---- 2b. lock step at BASELINE's full sizes -------------------------------------
@pytest.mark.parametrize("sim,worlds,steps,agents", [
    ("escape_room", 4096, 300, 2),          # BASELINE configs[1]
    ("escape_room_phys", 8192, 120, 2),     # configs[2]
    ("hideseek", 8192, 120, 5),             # configs[3], one GPU's share
])
def test_full_size_lockstep(built, sim, worlds, steps, agents):
    """The bench workloads themselves, every dumped column bit for bit against
    the reference CPU backend on all host cores (size-independent properties
    further down cover what a comparison cannot: id uniqueness etc.)."""
    _need_ref(sim)
    probs, step = run_pair(sim, worlds, steps, flags=200, check_every=20,
                           actions=_escape_actions(77, grab=sim != "escape_room",
                                                   agents=agents),
                           check_init=False, ref_workers=0)
    assert not probs, (step, probs[:3])


# ---- 2c. what an UNCHANGED simulator gets ----------------------------------------
@pytest.mark.parametrize("sim,worlds,steps,agents,denom", [
    ("escape_room", 4096, 200, 2, 200),         # BASELINE configs[1], its size
    ("escape_room", 333, 120, 2, 25),
    ("escape_room_phys", 2048, 120, 2, 60),
    ("hideseek", 2048, 120, 5, 80),
])
def test_portable_sources_lockstep(built, sim, worlds, steps, agents, denom):
    """lib<sim>_portable_hip.so: the bench simulators compiled for the GPU from
    their PORTABLE sources (-DSIM_PORTABLE: plain makeEntity / destroyEntity /
    findEntitiesWithinAABB with one lane per world, none of this backend's
    wave-cooperative extensions) -- what a reference simulator that links
    unchanged runs.  Bit for bit against the reference CPU backend, like the
    tuned builds."""
    _need_ref(sim)
    probs, step = run_pair(sim, worlds, steps, flags=denom, check_every=20,
                           actions=_escape_actions(31, grab=sim != "escape_room",
                                                   agents=agents),
                           check_init=False, ref_workers=0,
                           hip_sim=sim + "_portable")
    assert not probs, (step, probs[:3])


@pytest.mark.parametrize("seed", [11, 101])
@pytest.mark.parametrize("sim,agents,steps,denom", [
    ("escape_room", 2, 200, 60), ("escape_room_phys", 2, 120, 60),
    ("hideseek", 5, 120, 80), ("ball_pit", 0, 150, 50), ("sort_stress", 0, 80, 0)])
def test_lockstep_other_seeds(built, sim, agents, steps, denom, seed):
    """Every simulator again, 1024 worlds, other world seeds and action streams
    (this sweep is what found the candidate capacity of the LDS step kernel too
    small for a dense pile: ball_pit, seed 11)."""
    _need_ref(sim)
    probs, step = run_pair(
        sim, 777 if sim == "sort_stress" else 1024, steps, seed=seed,
        flags=denom, check_every=20,
        actions=_escape_actions(seed, grab=sim != "escape_room", agents=agents)
        if agents else None,
        check_init=False, ref_workers=0)
    assert not probs, (step, probs[:3])


# ---- 3. full-size properties -----------------------------------------------------
def _check_entity_columns(dump, archetypes):
    """ids unique across archetypes, none destroyed, worlds contiguous."""
    all_ids = []
    for arch in archetypes:
        rows, counts = dump[f"{arch}.Entity"]
        ents = rows.view(np.int32).reshape(-1, 2)   # gen, id
        assert (ents[:, 1] >= 0).all(), f"{arch}: destroyed row survived compaction"
        assert counts.sum() == len(ents)
        all_ids.append(ents[:, 1])
    ids = np.concatenate(all_ids)
    assert len(np.unique(ids)) == len(ids), "entity id handed out twice"


def test_escape_room_full_size_properties(built):
    """BASELINE config 2 size (4096 worlds) and config 3's world count (8192)."""
    for W in (4096, 8192):
        with Simulator(hip_lib_path("escape_room"), W, flags=100) as s:
            s.step(150)
            dump = s.dump_all()
            _check_entity_columns(dump, ["Agent", "PhysicsEntity", "DoorEntity",
                                         "ButtonEntity"])
            assert (dump["Agent.Entity"][1] == 2).all()
            assert (dump["PhysicsEntity.Entity"][1] == 18).all()
            assert (dump["DoorEntity.Entity"][1] == 3).all()
            assert (dump["ButtonEntity.Entity"][1] == 6).all()
            obs = s.read_tensor("self_obs")
            assert np.isfinite(obs).all()
            lidar = s.read_tensor("lidar")
            assert np.isfinite(lidar).all() and (lidar[..., 0] >= 0).all()


def test_escape_room_physics_full_size_properties(built):
    """BASELINE config 3 at its full size (8192 worlds): bodies stay finite and
    on or above the (infinite) floor plane; body counts per world constant; ids
    unique; two runs identical.  (Bodies may leave the arena: a grabbed cube
    can be dragged over a wall, exactly as on the CPU backend.)"""
    W = 8192
    dumps = []
    for _ in range(2):
        with Simulator(hip_lib_path("escape_room_phys"), W, flags=120) as s:
            rng = np.random.default_rng(9)
            for _step in range(40):
                a = np.stack([rng.integers(0, 4, (W, 2)),
                              rng.integers(0, 8, (W, 2)),
                              rng.integers(-2, 3, (W, 2)),
                              rng.integers(0, 2, (W, 2))], -1).astype(np.int32)
                s.write_tensor("action", a)
                s.step(1)
            dumps.append(s.dump_all())
    dump = dumps[0]
    assert not compare_columns(dump, dumps[1])
    _check_entity_columns(dump, ["Agent", "PhysicsEntity", "DoorEntity",
                                 "ButtonEntity"])
    assert (dump["Agent.Entity"][1] == 2).all()
    assert (dump["PhysicsEntity.Entity"][1] == 23).all()
    assert (dump["DoorEntity.Entity"][1] == 3).all()
    for arch in ("Agent", "PhysicsEntity"):
        pos = dump[f"{arch}.Position"][0].view(np.float32)
        assert np.isfinite(pos).all()
        assert (pos[:, 2] > -0.1).all(), "a body fell through the floor"
        assert (np.abs(pos[:, :2]) < 1e3).all()
        vel = dump[f"{arch}.Velocity"][0].view(np.float32)
        assert np.isfinite(vel).all()


def test_partition_invariance_on_device(built):
    """Worlds [0,64) stepped as one executor == two executors of 32 worlds with
    world_base 0 / 32 (what multi-GPU sharding relies on)."""
    steps = 80
    with Simulator(hip_lib_path("escape_room"), 64, flags=40) as whole:
        whole.step(steps)
        ref = whole.dump_all()
    parts = []
    for base in (0, 32):
        with Simulator(hip_lib_path("escape_room"), 32, flags=40, world_base=base) as s:
            s.step(steps)
            parts.append(s.dump_all())
    for col, (rows, counts) in ref.items():
        if col.endswith(".Entity") or col in ("Agent.OtherAgents",
                                              "DoorEntity.DoorProperties"):
            continue    # entity ids are executor-local
        cat_rows = np.concatenate([p[col][0] for p in parts])
        cat_counts = np.concatenate([p[col][1] for p in parts])
        assert np.array_equal(counts, cat_counts), col
        assert np.array_equal(rows, cat_rows), col


def test_determinism_run_to_run(built):
    """Two identical runs produce identical state (no scheduling dependence)."""
    dumps = []
    for _ in range(2):
        with Simulator(hip_lib_path("sort_stress"), 3000, seed=3) as s:
            s.step(40)
            dumps.append(s.dump_all())
    assert not compare_columns(dumps[0], dumps[1])


# ---- exported tensors as PyTorch-ROCm tensors -------------------------------------
def test_torch_tensor_bridge(built):
    import torch
    from madrona_amd.tensor import to_torch

    with Simulator(hip_lib_path("cartpole"), 256) as s:
        state = to_torch(s, "state")
        action = to_torch(s, "action")
        assert state.is_cuda and state.shape == (256, 4) and state.dtype == torch.float32
        before = state.clone()
        action.fill_(1)                       # written from torch, read by the next step
        torch.cuda.synchronize()
        s.step(1)
        assert not torch.equal(before, state)  # zero-copy view sees the new state
        assert np.array_equal(state.cpu().numpy(), s.read_tensor("state"))
        # pushing right (action 1) accelerates every cart to the right
        assert (state[:, 1] > before[:, 1]).all()


def test_stream_ordered_stepping_matches_synchronous(built):
    """step_async (MWCudaExecutor::runAsync on the executor's own stream) and the
    stream-ordered ShardedSimulator path (pack ordered after the replay by
    events, no host round trip per step) give what synchronous steps give."""
    import torch
    from madrona_amd.distributed import ShardedSimulator, shard_for

    W, steps = 256, 40
    names = ["self_obs", "lidar", "reward", "done"]
    with Simulator(hip_lib_path("escape_room"), W, flags=20) as ref:
        ref.step(steps)
        expect = {n: ref.read_tensor(n) for n in names}
        expect_cols = ref.dump_all()

    with Simulator(hip_lib_path("escape_room"), W, flags=20) as s:
        s.step_async(steps)
        s.sync()
        assert not compare_columns(expect_cols, s.dump_all())

    sharded = ShardedSimulator(
        lambda n, base: Simulator(hip_lib_path("escape_room"), n, flags=20,
                                  world_base=base),
        shard_for(0, 1, total_worlds=W), names)
    out = None
    for _ in range(steps):
        out = sharded.step(1)
    torch.cuda.synchronize()
    sharded.sync()
    for n in names:
        assert np.array_equal(out[n].cpu().numpy().reshape(expect[n].shape).view(np.uint8),
                              expect[n].view(np.uint8)), n
    sharded.close()


@pytest.mark.parametrize("sim,div", [("escape_room", 2), ("escape_room", 7),
                                     ("escape_room_phys", 2), ("sort_stress", 2)])
def test_tables_grow_in_place(built, monkeypatch, sim, div):
    """Table growth: tables start with 1/div of the rows the
    simulator declared; the executor maps more memory behind their columns
    between replays (hipMemAddressReserve + 2 MiB chunks, addresses unchanged)
    and rebuilds its launch graphs.  Results stay bit-identical to the
    reference, and exported tensors taken before the growth stay valid."""
    import ctypes as C
    from madrona_amd.simlib import runtime_lib
    _need_ref(sim)
    monkeypatch.setenv("MADRONA_MWHIP_INITIAL_CAPACITY_DIV", str(div))

    # (div = 2: exactly what the constructors create; div = 7: less, so the
    # tables already grow while the worlds are being constructed)
    worlds, steps = 200, 150
    kw = dict(flags=12) if sim.startswith("escape_room") else {}
    with Simulator(ref_lib_path(sim), worlds, seed=5, num_workers=1, **kw) as ref, \
            Simulator(hip_lib_path(sim), worlds, seed=5, **kw) as hip:
        rt = runtime_lib()
        rt.mwhip_num_table_growths.restype = C.c_uint32
        rt.mwhip_num_table_growths.argtypes = [C.c_void_p]
        first_ptrs = {n: hip.tensor_ptr(n) for n in hip.tensor_names}
        grown_at_start = rt.mwhip_num_table_growths(hip.hip_exec())
        feed = _escape_actions(3, grab=sim == "escape_room_phys") \
            if sim.startswith("escape_room") else None
        for s in range(1, steps + 1):
            if feed is not None:
                feed(ref, hip, s)
            ref.step(1)
            hip.step(1)
            if s % 10 == 0 or s == steps:
                probs = compare_columns(ref.dump_all(), hip.dump_all())
                assert not probs, (s, probs[:3])
        grown = rt.mwhip_num_table_growths(hip.hip_exec())
        assert grown > 0 and grown >= grown_at_start, "nothing grew"
        assert first_ptrs == {n: hip.tensor_ptr(n) for n in hip.tensor_names}
        for name in ref.tensor_names:
            assert np.array_equal(ref.read_tensor(name).view(np.uint8),
                                  hip.read_tensor(name).view(np.uint8)), name


def test_tables_grow_between_replays(built, monkeypatch):
    """Growth while stepping (not only after construction): sort_stress in its
    ramp-up mode -- every world starts with one item and creates up to six per
    step until it holds 40 -- with tables mapped for a quarter of what the
    simulator declared.  Everything stays bit-identical to the reference."""
    import ctypes as C
    from madrona_amd.simlib import runtime_lib
    _need_ref("sort_stress")
    monkeypatch.setenv("MADRONA_MWHIP_INITIAL_CAPACITY_DIV", "4")
    W, steps = 300, 60
    with Simulator(ref_lib_path("sort_stress"), W, seed=7, num_workers=1, flags=2) as r, \
            Simulator(hip_lib_path("sort_stress"), W, seed=7, flags=2) as h:
        rt = runtime_lib()
        rt.mwhip_num_table_growths.restype = C.c_uint32
        rt.mwhip_num_table_growths.argtypes = [C.c_void_p]
        grown_at_start = rt.mwhip_num_table_growths(h.hip_exec())
        for s in range(1, steps + 1):
            r.step(1)
            h.step(1)
            if s % 5 == 0:
                probs = compare_columns(r.dump_all(), h.dump_all())
                assert not probs, (s, probs[:3])
        assert rt.mwhip_num_table_growths(h.hip_exec()) >= grown_at_start + 2


def test_tables_grow_under_a_queue_of_replays(built, monkeypatch):
    """Growth has to keep ahead of replays that are queued without being waited
    for (MWCudaExecutor::runAsync): sort_stress ramping up from 1 to 40 items
    per world with tables mapped for a sixteenth of what the simulator
    declared, stepped through mwhip_run_async only.  The very first step
    already outruns what is mapped: the appending threads post their request in
    the mailbox and wait while the executor's service thread maps more memory
    behind the columns of the running kernel (on-demand growth, reference
    src/mw/device/memory.cpp:27-121); between replays tables are sized by the
    per-step high-water mark.  Results stay bit-identical."""
    import ctypes as C
    from madrona_amd.simlib import runtime_lib
    _need_ref("sort_stress")
    monkeypatch.setenv("MADRONA_MWHIP_INITIAL_CAPACITY_DIV", "16")
    W, steps = 300, 60
    with Simulator(ref_lib_path("sort_stress"), W, seed=7, num_workers=1, flags=2) as r, \
            Simulator(hip_lib_path("sort_stress"), W, seed=7, flags=2) as h:
        rt = runtime_lib()
        rt.mwhip_num_table_growths.restype = C.c_uint32
        rt.mwhip_num_table_growths.argtypes = [C.c_void_p]
        grown_at_start = rt.mwhip_num_table_growths(h.hip_exec())
        for chunk in range(steps // 20):
            r.step(20)
            h.step_async(20)        # 20 replays queued back to back
            h.sync()
            probs = compare_columns(r.dump_all(), h.dump_all())
            assert not probs, (chunk, probs[:3])
        assert rt.mwhip_num_table_growths(h.hip_exec()) >= grown_at_start + 3


def test_entity_store_and_scratch_region_grow(built, monkeypatch):
    """The entity store and the Context::tmpAlloc region live in reserved address
    space like the tables.  Started at a sixty-fourth of their
    default sizes: the world constructors already outrun the id store (it grows
    between the constructor passes), the first steps outrun the 1 MiB scratch
    region and the ids (requests through the mailbox while the replay runs).
    Results stay bit-identical to the reference; worlds that start empty take
    run-time id blocks from memory mapped on demand."""
    import ctypes as C
    from madrona_amd.simlib import runtime_lib
    _need_ref("sort_stress")
    monkeypatch.setenv("MADRONA_MWHIP_INITIAL_ID_CAPACITY_DIV", "64")
    monkeypatch.setenv("MADRONA_MWHIP_TMP_MB", "1")
    rt = runtime_lib()
    rt.mwhip_num_table_growths.restype = C.c_uint32
    rt.mwhip_num_table_growths.argtypes = [C.c_void_p]

    # populated worlds: ids bit for bit (one 256-byte scratch block per world and
    # step: 6000 worlds ask for 1.5 MiB)
    W, steps = 6000, 24
    with Simulator(ref_lib_path("sort_stress"), W, seed=3, num_workers=1) as r, \
            Simulator(hip_lib_path("sort_stress"), W, seed=3) as h:
        for chunk in range(steps // 8):
            r.step(8)
            h.step_async(8)
            h.sync()
            probs = compare_columns(r.dump_all(), h.dump_all())
            assert not probs, (chunk, probs[:3])
        assert rt.mwhip_num_table_growths(h.hip_exec()) >= 2

    # worlds that start empty: run-time id blocks (ids are a renaming, above)
    W = 255
    with Simulator(ref_lib_path("sort_stress"), W, seed=7, num_workers=1, flags=1) as r, \
            Simulator(hip_lib_path("sort_stress"), W, seed=7, flags=1) as h:
        r.step(30)
        h.step(30)
        ref, hip = r.dump_all(), h.dump_all()
        ref_ids = ref.pop("Item.Entity")[0].view(np.int32).reshape(-1, 2)
        hip_ids = hip.pop("Item.Entity")[0].view(np.int32).reshape(-1, 2)
        assert not compare_columns(ref, hip)
        assert np.array_equal(ref_ids[:, 0], hip_ids[:, 0])
        pairs = np.unique(np.stack([ref_ids[:, 1], hip_ids[:, 1]], 1), axis=0)
        assert len(np.unique(pairs[:, 0])) == len(pairs) == len(np.unique(pairs[:, 1]))
        assert rt.mwhip_num_table_growths(h.hip_exec()) >= 1


@pytest.mark.parametrize("sim,worlds", [("escape_room", 300), ("escape_room_phys", 64)])
def test_input_ring_feeds_actions(built, sim, worlds):
    """mwhip_set_input_ring: replays queued back to back take their actions from
    a device-resident ring, slot (replays completed) % slots -- the same run as
    writing the action tensor before every step."""
    import torch
    from madrona_amd.tensor import to_torch
    slots, steps = 5, 23
    rng = np.random.default_rng(4)
    ring = np.stack([np.stack([rng.integers(0, 4, (worlds, 2)), rng.integers(0, 8, (worlds, 2)),
                               rng.integers(-2, 3, (worlds, 2)),
                               rng.integers(0, 2, (worlds, 2))], -1)
                     for _ in range(slots)]).astype(np.int32)
    with Simulator(hip_lib_path(sim), worlds, seed=3, flags=9) as a, \
            Simulator(hip_lib_path(sim), worlds, seed=3, flags=9) as b:
        dev = torch.from_numpy(ring).cuda()
        a.set_input_ring("action", dev.data_ptr(), slots)
        a.step_async(steps)
        a.sync()
        for step in range(steps):
            b.write_tensor("action", ring[step % slots])
            b.step(1)
        assert not compare_columns(a.dump_all(), b.dump_all())
        # (the simulators clear the actions of worlds they reset)
        assert np.array_equal(a.read_tensor("action"), b.read_tensor("action"))
        # removing the ring: the tensor is the caller's again
        a.set_input_ring("action", 0, 1)
        for s in (a, b):
            s.write_tensor("action", ring[0])
            s.step(3)
        assert not compare_columns(a.dump_all(), b.dump_all())
        del dev, to_torch


@pytest.mark.parametrize("worlds,extra,steps", [(96, 0, 60), (24, 60, 40), (16, 140, 30)])
def test_ball_pit_wave_box_queries(built, monkeypatch, worlds, extra, steps):
    """PhysicsSystem::findFirstEntitiesWithinAABBsWave against the reference's
    findEntitiesWithinAABB: ball_pit asks four boxes per world and step (the CPU
    build through the tree walk, the GPU build through the wavefront) and keeps
    a running digest of the answers -- on worlds with spheres, a two-primitive
    object, and in crowd mode more BVH leaves than lanes (79 / 159 bodies)."""
    _need_ref("ball_pit")
    # (crowds as in test_ball_pit_crowd: candidate / contact budgets of the HBM
    # step kernel, no resets for the big one)
    monkeypatch.setenv("MADRONA_MWHIP_MAX_CANDIDATES_PER_WORLD", "2048")
    monkeypatch.setenv("MADRONA_MWHIP_MAX_CONTACTS_PER_WORLD", "1024")
    flags = (extra << 16) | (25 if extra <= 60 else 0)
    # (seed and thread count of the other ball_pit tests: the reference's debug
    # asserts in its sphere-hull path abort on some other trajectories)
    with Simulator(ref_lib_path("ball_pit"), worlds, seed=5, flags=flags,
                   num_workers=1) as ref, \
            Simulator(hip_lib_path("ball_pit"), worlds, seed=5, flags=flags) as hip:
        for step in range(1, steps + 1):
            ref.step(1)
            hip.step(1)
            if step % 10 == 0 or step == steps:
                r, h = ref.read_tensor("query_probe"), hip.read_tensor("query_probe")
                assert np.array_equal(r, h), (step, np.flatnonzero((r != h).any(1))[:5])
        assert not compare_columns(ref.dump_all(), hip.dump_all())
        found = hip.read_tensor("query_probe")[:, 1]
        # (the probes do find bodies, in every world)
        assert (found > steps).all(), found.min()


def test_same_dependency_siblings_that_clash_keep_their_order(built):
    """Nodes that named the same dependencies share one launch -- unless their
    signatures clash: sort_stress with three siblings behind one
    node, the second reading the Quad the first writes (right only in
    registration order, which is what the reference's executors run), the third
    on a column of its own.  Lock step with the reference, and the launch list:
    the writer alone, the other two side by side."""
    _need_ref("sort_stress")
    probs, step = run_pair("sort_stress", 300, 25, seed=7, flags=16, check_every=5)
    assert not probs, (step, probs[:3])
    with Simulator(hip_lib_path("sort_stress"), 300, seed=7, flags=16) as s:
        s.step(2)
        names = [k["name"] for k in s.profile(2)]
    assert "sortstress::siblingWriteSystem" in names, names
    groups = [n for n in names if n.startswith("group[")]
    assert groups == ["group[sortstress::siblingReadSystem | "
                      "sortstress::siblingOtherSystem]"], names
