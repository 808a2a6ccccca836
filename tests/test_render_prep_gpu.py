"""GPU: the render-prep ECS systems (reference
src/render/ecs_system.cpp) on sims/render_prep -- instance / view / light
records, Morton codes, and the sort chains of RenderingSystem::setupTasks,
two of them SortArchetypeNode over non-WorldID keys.

Oracle: the reference itself, stepped in lock step.  Its CPU mode writes
instance and view records into RenderECSBridge buffers (owned here by the
simulator's manager) instead of into the render entities' rows, and its CPU
sort by a non-world key is not an oracle for order, so:
  * simulator-side columns, entity ids and the light table: bit for bit;
  * instance records: per world, the multiset of the 64-byte records the
    reference appended == the multiset of the HIP table's InstanceData rows;
  * Morton codes: per-world multisets against the reference's column, and
    recomputed here from the record's position bits;
  * order (the point of the sort chains): checked on the HIP tables directly --
    grouped by world, Morton-ascending inside a world, views in output-slot order;
  * view records: position / rotation / zNear / world exact, the two fov scales
    to 1e-6 (tanf of the host libm vs the device's)."""
import ctypes as C

import numpy as np
import pytest

from madrona_amd.simlib import Simulator, hip_lib_path, ref_lib_path
from parity_utils import compare_columns

pytestmark = pytest.mark.gpu

EXACT = ["Mover.Position", "Mover.Rotation", "Mover.Renderable",
         "Mover.MaterialOverride", "Mover.ColorOverride", "Viewer.Position",
         "Lamp.Position"]
# LightDesc { bool type, castShadow; <2 pad>; vec3 position, direction; float
# cutoff, intensity; bool active; <3 pad> }: padding is whatever the stack held
LIGHT_BYTES = [b for b in range(40) if b not in (2, 3, 37, 38, 39)]

ROOT_AABBS = np.array([
    [-0.5, -0.5, -0.5, 0.5, 0.5, 0.5], [-1, -0.25, 0, 1, 0.25, 2],
    [-0.75, -0.75, -0.1, 0.75, 0.75, 0.1], [0, 0, 0, 1.5, 1, 0.5]], np.float32)


def _bridge(sim, kind, worlds):
    """Records the reference appended during the last step, grouped by world."""
    sim.lib.render_prep_bridge_records.restype = C.c_int64
    sim.lib.render_prep_bridge_records.argtypes = [C.c_int32, C.c_void_p, C.c_void_p,
                                                   C.c_uint64]
    size = 64 if kind == 0 else 48
    cap = worlds * 128
    rec = np.zeros((cap, size), np.uint8)
    keys = np.zeros(cap, np.uint64)
    n = sim.lib.render_prep_bridge_records(kind, rec.ctypes.data, keys.ctypes.data, cap)
    assert n >= 0
    world = (keys[:n] >> np.uint64(32)).astype(np.int64)
    return [rec[:n][world == w] for w in range(worlds)]


def _split(rows, counts):
    out, at = [], 0
    for c in counts:
        out.append(rows[at:at + c])
        at += c
    return out


def _sorted_rows(rows):
    if len(rows) == 0:
        return rows
    return rows[np.lexsort(rows.T[::-1])]


def _spread10(v):
    v = np.where(v == 1024, 1023, v).astype(np.uint32)
    v = (v | (v << 16)) & 0x030000FF
    v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3
    v = (v | (v << 2)) & 0x09249249
    return v


def _morton(pos_bits):
    """reference ecs_system.cpp:51-83 on the raw float bits of x, y, z"""
    x, y, z = (_spread10(pos_bits[:, i]) for i in range(3))
    return (z << 2) | (y << 1) | x


@pytest.mark.parametrize("worlds,steps", [(1, 40), (37, 60), (900, 25), (8192, 10)])
def test_render_prep_lockstep(built, worlds, steps):
    import os
    if not os.path.exists(ref_lib_path("render_prep")):
        pytest.skip("oracle/_ref missing on this box")
    with Simulator(ref_lib_path("render_prep"), worlds, seed=3, num_workers=1) as ref, \
            Simulator(hip_lib_path("render_prep"), worlds, seed=3) as hip:
        for step in range(1, steps + 1):
            ref.step(1)
            hip.step(1)
            if worlds > 100 and step % 5 != 0 and step != steps:
                continue
            rd, hd = ref.dump_all(), hip.dump_all()
            probs = compare_columns({k: rd[k] for k in EXACT}, hd)
            assert not probs, (step, probs[:3])
            assert np.array_equal(rd["Light.LightDesc"][1], hd["Light.LightDesc"][1])
            assert np.array_equal(rd["Light.LightDesc"][0][:, LIGHT_BYTES],
                                  hd["Light.LightDesc"][0][:, LIGHT_BYTES]), step
            roster = hip.read_tensor("roster")
            assert np.array_equal(ref.read_tensor("roster"), roster)

            # ---- instances ----
            inst_rows, inst_counts = hd["Renderable.InstanceData"]
            assert np.array_equal(inst_counts, roster[:, 1]), step   # drawn movers
            hip_inst = _split(inst_rows, inst_counts)
            ref_inst = _bridge(ref, 0, worlds)
            mort_rows, mort_counts = hd["Renderable.MortonCode"]
            assert np.array_equal(mort_counts, inst_counts)
            hip_mort = _split(mort_rows.view(np.uint32).ravel(), mort_counts)
            ref_mort = _split(rd["Renderable.MortonCode"][0].view(np.uint32).ravel(),
                              rd["Renderable.MortonCode"][1])
            for w in range(worlds):
                # (56 bytes of fields: position, rotation, scale, matID, objectID,
                # worldIDX, color; the last 8 are alignment padding)
                assert np.array_equal(_sorted_rows(hip_inst[w][:, :56]),
                                      _sorted_rows(ref_inst[w][:, :56])), (step, w)
                words = hip_inst[w].view(np.uint32).reshape(-1, 16)
                assert (words[:, 12].view(np.int32) == w).all()          # worldIDX
                # sorted by Morton code inside the world, code == f(position bits)
                assert (hip_mort[w][:-1] <= hip_mort[w][1:]).all(), (step, w)
                assert np.array_equal(hip_mort[w], _morton(words[:, 0:3])), (step, w)
                assert np.array_equal(np.sort(hip_mort[w]), np.sort(ref_mort[w])), (step, w)

            # ---- views ----
            cam_rows, cam_counts = hd["Camera.PerspectiveCameraData"]
            assert (cam_counts == 2).all()
            idx = hd["Camera.RenderOutputIndex"][0].view(np.uint32).ravel()
            assert np.array_equal(idx, np.arange(2 * worlds))            # output slots
            hip_cam = cam_rows.view(np.float32).reshape(-1, 12)
            ref_cam = np.concatenate(_bridge(ref, 1, worlds)).view(np.float32).reshape(-1, 12)
            assert np.array_equal(hip_cam[:, :7].view(np.uint32), ref_cam[:, :7].view(np.uint32))
            assert np.array_equal(hip_cam[:, 9:11].view(np.uint32), ref_cam[:, 9:11].view(np.uint32))
            assert np.allclose(hip_cam[:, 7:9], ref_cam[:, 7:9], rtol=1e-6, atol=0)


def test_instance_table_without_the_reference_s_first_compaction(built, monkeypatch):
    """VERDICT r04 #9 / DESIGN section 8: the reference compacts the instance
    table BEFORE the Morton sort as well (src/render/ecs_system.cpp:550-553);
    this backend leaves that chain out.  The claim -- the table that leaves the
    chains is the same, because both sorts are stable and a compaction keeps the
    relative order of a world's rows -- checked row for row: the same simulator
    built with that node (MADRONA_MWHIP_RENDER_PRECOMPACT=1: the reference's
    graph) and without, movers destroyed, created, hidden and shown every step
    (rows marked dead in the table the Morton sort sees, new rows behind the
    sorted prefix), standing on nine grid points so that most of a world's
    instances tie on their Morton code.  Instance data, codes and their order
    must be identical every step."""
    worlds, flags = 300, 16 | 4     # ties + dense worlds (70..95 movers)
    monkeypatch.setenv("MADRONA_MWHIP_RENDER_PRECOMPACT", "1")
    with_node = Simulator(hip_lib_path("render_prep"), worlds, seed=11, flags=flags)
    monkeypatch.delenv("MADRONA_MWHIP_RENDER_PRECOMPACT")
    without = Simulator(hip_lib_path("render_prep"), worlds, seed=11, flags=flags)
    try:
        assert len(with_node.profile(1)) > len(without.profile(1))   # one more chain
        for step in range(1, 61):
            with_node.step(1)
            without.step(1)
            a, b = with_node.dump_all(), without.dump_all()
            for col in ("Renderable.InstanceData", "Renderable.MortonCode"):
                assert np.array_equal(a[col][1], b[col][1]), (step, col)
                assert np.array_equal(a[col][0], b[col][0]), (step, col)
            codes = _split(b["Renderable.MortonCode"][0].view(np.uint32).ravel(),
                           b["Renderable.MortonCode"][1])
        # the ties were there: far fewer distinct codes than instances
        assert sum(len(np.unique(c)) for c in codes) * 4 < sum(len(c) for c in codes)
    finally:
        with_node.close()
        without.close()


def test_render_prep_visual_overrides_and_tlbvh(built):
    """update_visual_properties = true (material / colour overrides reach the
    records, lights follow their carriers) with the ray caster configured: TLBVH
    leaf boxes, render targets per view."""
    worlds = 64
    flags = 1 | (16 << 8)       # overrides on, 16 x 16 outputs
    with Simulator(hip_lib_path("render_prep"), worlds, seed=9, flags=flags) as hip:
        for step in range(30):
            hip.step(1)
            d = hip.dump_all()
            ent = d["Renderable.Entity"][0].view(np.int32).reshape(-1, 2)
            inst = d["Renderable.InstanceData"][0].view(np.uint32).reshape(-1, 16)
            row_of = {int(i): r for r, i in enumerate(ent[:, 1])}
            owner = d["Mover.Renderable"][0].view(np.int32).reshape(-1, 2)
            mat = d["Mover.MaterialOverride"][0].view(np.int32).ravel()
            col = d["Mover.ColorOverride"][0].view(np.uint32).ravel()
            seen = 0
            for m in range(len(owner)):
                if owner[m, 1] < 0:
                    continue        # hidden mover
                r = row_of[int(owner[m, 1])]
                assert inst[r, 10].view(np.int32) == mat[m] and inst[r, 13] == col[m]
                seen += 1
            assert seen == len(inst)

            # TLBVH leaf box = root box of the object, transformed (Arvo)
            f = inst.view(np.float32)
            pos, q, s = f[:, 0:3], f[:, 3:7], f[:, 7:10]
            obj = inst[:, 11].view(np.int32)
            w_, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
            R = np.stack([
                np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w_ * z), 2 * (x * z + w_ * y)], -1),
                np.stack([2 * (x * y + w_ * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w_ * x)], -1),
                np.stack([2 * (x * z - w_ * y), 2 * (y * z + w_ * x), 1 - 2 * (x * x + y * y)], -1),
            ], 1) * s[:, None, :]
            lo, hi = ROOT_AABBS[obj][:, :3], ROOT_AABBS[obj][:, 3:]
            e = R * lo[:, None, :]
            g = R * hi[:, None, :]
            want_min = pos + np.minimum(e, g).sum(-1)
            want_max = pos + np.maximum(e, g).sum(-1)
            box = d["Renderable.TLBVHNode"][0].view(np.float32).reshape(-1, 8)
            assert np.allclose(box[:, 0:3], want_min, rtol=1e-5, atol=1e-5)
            assert np.allclose(box[:, 3:6], want_max, rtol=1e-5, atol=1e-5)

            # every view owns a render target entity
            refs = d["Camera.RenderOutputRef"][0].view(np.int32).reshape(-1, 2)
            assert (refs[:, 1] >= 0).all() and len(np.unique(refs[:, 1])) == 2 * worlds
