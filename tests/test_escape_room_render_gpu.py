"""GPU: BASELINE config 5 pinned at the size it is measured at.

(1) `sims/escape_room_render` -- the Escape Room with rigid-body physics AND the
    render-prep systems in one task graph (resets re-creating render entities,
    34 instances per world, the sort chains of RenderingSystem::setupTasks behind
    the physics nodes) -- in lock step with the reference CPU backend
    (oracle/_ref/libescape_room_render_ref.so: reference src/physics/*.cpp +
    src/render/ecs_system.cpp:486-597 compiled where they lie), at 64 / 1024 /
    8192 worlds.  Simulator columns, entity ids, render-entity handles and the
    light table bit for bit; instance records as per-world multisets of their 56
    payload bytes (the reference's CPU mode appends them to RenderECSBridge
    buffers in arrival order, its Morton sort is not an order oracle: SURVEY
    a16); order checked on the HIP tables (grouped by world, Morton-ascending,
    views in output-slot order); view records exact except the two fov scales
    (tanf of two libms, 1e-6).

(2) The config-5 render pass at full size: 8192 worlds x 2 agents x 64 x 64,
    all 16384 views against the reference's ray caster compiled for the host
    (reference src/mw/device/bvh_raycast.cpp:940-1029 through
    oracle/ref_shims/raycast_ref_shim.cpp)."""
import ctypes as C
import os

import numpy as np
import pytest

from madrona_amd.simlib import Simulator, hip_lib_path, ref_lib_path
from parity_utils import compare_columns
from raycast_utils import REF_LIB
from test_raycast_gpu import _compare, _reference_images, _sim_geometry

pytestmark = pytest.mark.gpu

# LightDesc { bool type, castShadow; <2 pad>; vec3 position, direction; float
# cutoff, intensity; bool active; <3 pad> }: padding is whatever the stack held
LIGHT_BYTES = [b for b in range(40) if b not in (2, 3, 37, 38, 39)]
INSTANCES_PER_WORLD = 34   # floor + 4 borders + 2 agents + 3 x (2 walls + door + 4 cubes + 2 buttons)


def _actions(rng, worlds):
    return np.stack([rng.integers(0, 4, (worlds, 2)), rng.integers(0, 8, (worlds, 2)),
                     rng.integers(-2, 3, (worlds, 2)), rng.integers(0, 2, (worlds, 2))],
                    -1).astype(np.int32)


def _bridge(sim, kind, worlds):
    """Records the reference appended during the last step + their world."""
    f = sim.lib.render_prep_bridge_records
    f.restype = C.c_int64
    f.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64]
    size = 64 if kind == 0 else 48
    cap = worlds * 64
    rec = np.zeros((cap, size), np.uint8)
    keys = np.zeros(cap, np.uint64)
    n = f(kind, rec.ctypes.data, keys.ctypes.data, cap)
    assert n >= 0
    return rec[:n], (keys[:n] >> np.uint64(32)).astype(np.int64), \
        (keys[:n] & np.uint64(0xFFFFFFFF)).astype(np.int64)


def _canonical(rows56, world):
    """rows sorted by (world, payload): equal arrays <=> equal per-world multisets"""
    words = np.ascontiguousarray(rows56).view(np.uint64).reshape(-1, 7)
    order = np.lexsort([words[:, k] for k in range(6, -1, -1)] + [world])
    return words[order], world[order]


def _spread10(v):
    v = np.where(v == 1024, 1023, v).astype(np.uint32)
    v = (v | (v << 16)) & 0x030000FF
    v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3
    v = (v | (v << 2)) & 0x09249249
    return v


def _check_render_side(ref, hip, rd, hd, worlds, step):
    # ---- handles: the render entities' ids are the CPU backend's ----
    for col in ("PhysicsEntity.Renderable", "ButtonEntity.Renderable", "Agent.Renderable"):
        assert np.array_equal(rd[col][0], hd[col][0]), (step, col)
    cam_r, cam_h = rd["Agent.RenderCamera"][0], hd["Agent.RenderCamera"][0]
    keep = [b for b in range(28) if not 8 <= b < 12]
    assert np.array_equal(cam_r[:, keep], cam_h[:, keep]), step
    assert np.allclose(cam_r[:, 8:12].copy().view(np.float32),
                       cam_h[:, 8:12].copy().view(np.float32), rtol=1e-6, atol=0)
    assert np.array_equal(rd["Light.LightDesc"][1], hd["Light.LightDesc"][1])
    assert np.array_equal(rd["Light.LightDesc"][0][:, LIGHT_BYTES],
                          hd["Light.LightDesc"][0][:, LIGHT_BYTES]), step

    # ---- instance records: per-world multisets ----
    inst, counts = hd["Renderable.InstanceData"]
    assert (counts == INSTANCES_PER_WORLD).all(), step
    hip_world = np.repeat(np.arange(worlds), counts)
    ref_inst, ref_world, _ = _bridge(ref, 0, worlds)
    assert np.array_equal(np.bincount(ref_world, minlength=worlds), counts), step
    hw, hww = _canonical(inst[:, :56], hip_world)
    rw, rww = _canonical(ref_inst[:, :56], ref_world)
    assert np.array_equal(hww, rww) and np.array_equal(hw, rw), step
    words = inst.view(np.uint32).reshape(-1, 16)
    assert np.array_equal(words[:, 12].view(np.int32), hip_world), step     # worldIDX

    # ---- order on the HIP table: Morton-ascending inside a world, code == f(position bits)
    # (reference ecs_system.cpp:51-83) ----
    mort, mcounts = hd["Renderable.MortonCode"]
    assert np.array_equal(mcounts, counts)
    code = mort.view(np.uint32).ravel()
    x, y, z = (_spread10(words[:, i]) for i in range(3))
    assert np.array_equal(code, (z << 2) | (y << 1) | x), step
    same_world = hip_world[1:] == hip_world[:-1]
    assert (code[1:][same_world] >= code[:-1][same_world]).all(), step

    # ---- views: output-slot order, records exact but for the fov scales ----
    cams, ccounts = hd["Camera.PerspectiveCameraData"]
    assert (ccounts == 2).all()
    idx = hd["Camera.RenderOutputIndex"][0].view(np.uint32).ravel()
    assert np.array_equal(idx, np.arange(2 * worlds)), step
    ref_views, view_world, view_ent = _bridge(ref, 1, worlds)
    assert len(ref_views) == 2 * worlds
    # arrival order of a multi-threaded reference run is not the table's: by
    # (world, camera entity id) -- a world's agents are created in id order
    order = np.lexsort([view_ent, view_world])
    hip_cam = cams.view(np.float32).reshape(-1, 12)
    ref_cam = ref_views[order].view(np.float32).reshape(-1, 12)
    assert np.array_equal(hip_cam[:, :7].view(np.uint32), ref_cam[:, :7].view(np.uint32)), step
    assert np.array_equal(hip_cam[:, 9:11].view(np.uint32),
                          ref_cam[:, 9:11].view(np.uint32)), step
    assert np.allclose(hip_cam[:, 7:9], ref_cam[:, 7:9], rtol=1e-6, atol=0)


@pytest.mark.parametrize("worlds,steps,denom", [(64, 120, 20), (1024, 60, 50),
                                                (8192, 40, 200)])
def test_escape_room_render_lockstep(built, worlds, steps, denom):
    if not os.path.exists(ref_lib_path("escape_room_render")):
        pytest.skip("oracle/_ref missing on this box")
    rng = np.random.default_rng(worlds)
    every = 1 if worlds <= 64 else 10
    with Simulator(ref_lib_path("escape_room_render"), worlds, seed=4, flags=denom,
                   num_workers=1 if worlds <= 64 else 0) as ref, \
            Simulator(hip_lib_path("escape_room_render"), worlds, seed=4,
                      flags=denom | (1 << 26)) as hip:    # (no render-target entities)
        for step in range(1, steps + 1):
            act = _actions(rng, worlds)
            ref.write_tensor("action", act)
            hip.write_tensor("action", act)
            ref.step(1)
            hip.step(1)
            if step % every != 0 and step != steps:
                continue
            rd, hd = ref.dump_all(), hip.dump_all()
            special = ("Light.LightDesc", "Agent.RenderCamera")
            probs = compare_columns({k: v for k, v in rd.items() if k not in special}, hd)
            assert not probs, (step, probs[:3])
            _check_render_side(ref, hip, rd, hd, worlds, step)
        for name in ref.tensor_names:
            assert np.array_equal(ref.read_tensor(name).view(np.uint8),
                                  hip.read_tensor(name).view(np.uint8)), name


def test_config5_render_pass_full_size(built):
    """BASELINE config 5 as bench.py runs it: 8192 worlds, two 64 x 64 RGB-D views
    each, every one of the 16384 views against the reference's ray caster."""
    if not os.path.exists(REF_LIB):
        pytest.skip("oracle/_ref/libraycast_ref.so missing on this box")
    worlds, res = 8192, 64
    with Simulator(hip_lib_path("escape_room_render"), worlds, seed=5,
                   flags=200 | (res << 16)) as hip:
        geo = _sim_geometry(hip)
        rng = np.random.default_rng(5)
        for step in range(12):
            hip.write_tensor("action", _actions(rng, worlds))
            hip.step(1)
        hip.render()
        d = hip.dump_all()
        assert (d["Renderable.InstanceData"][1] == INSTANCES_PER_WORLD).all()
        ref_rgb, ref_depth = _reference_images(geo, worlds, d, res, True,
                                               threads=os.cpu_count() or 1)
        hip_rgb, hip_depth = hip.read_tensor("rgb"), hip.read_tensor("depth")
        assert hip_depth.shape == (2 * worlds, res, res)
        # (67 M pixels; the runs show 1 depth beyond 1e-5 and 1 colour off by more
        # than a step: a ray grazing a triangle edge)
        _compare(hip_rgb, hip_depth, ref_rgb, ref_depth, True, ("config5", worlds, res),
                 flips=4, offs=4)


def test_input_ring_ignores_render_replays(built):
    """mwhip_set_input_ring with render-graph replays queued between the steps
    (config 5's loop: step, render, step, render ...): the k-th STEP takes slot
    k % slots -- render replays neither read nor advance the ring (ADVICE r3:
    the ring used to index by replays of any graph, so every render skipped a
    slot; an even number of slots then repeated half of the ring)."""
    import torch
    worlds, res, slots, steps = 48, 16, 4, 11
    rng = np.random.default_rng(11)
    ring = np.stack([_actions(rng, worlds) for _ in range(slots)])
    flags = 9 | (res << 16)
    with Simulator(hip_lib_path("escape_room_render"), worlds, seed=6, flags=flags) as a, \
            Simulator(hip_lib_path("escape_room_render"), worlds, seed=6, flags=flags) as b:
        dev = torch.from_numpy(ring).cuda()
        a.render()      # (builds the render graph; a render before the ring is set)
        b.render()
        a.set_input_ring("action", dev.data_ptr(), slots)
        render_graph = a.render_graph()
        for step in range(steps):
            a.step_async(1)
            a.step_async(1 + step % 2, graph=render_graph)
        a.sync()
        for step in range(steps):
            b.write_tensor("action", ring[step % slots])
            b.step(1)
        b.render()
        assert not compare_columns(a.dump_all(), b.dump_all())
        assert np.array_equal(a.read_tensor("action"), b.read_tensor("action"))
        assert np.array_equal(a.read_tensor("rgb"), b.read_tensor("rgb"))
        assert np.array_equal(a.read_tensor("depth"), b.read_tensor("depth"))
        del dev
