"""CPU: pins the ray-caster oracle (the reference's bvh_raycast.cpp compiled for
the host, oracle/ref_shims/raycast_ref_shim.cpp) against answers known in closed
form: the shim's own parts -- CUDA keywords defined away, the QBVH builder, the
per-pixel driver -- must not change what the reference computes."""
import os

import numpy as np
import pytest

from raycast_utils import (INSTANCE_DT, LIGHT_DT, REF_LIB, VIEW_DT, cube_geometry,
                           primary_rays, ref_render)

pytestmark = pytest.mark.skipif(not os.path.exists(REF_LIB),
                                reason="oracle/_ref/libraycast_ref.so not built")


def _view(fov_scale=1.0):
    v = np.zeros(1, VIEW_DT)
    v["rotation"] = [1, 0, 0, 0]
    v["xScale"], v["yScale"] = fov_scale, -fov_scale
    return v


def _instance(pos, scale=(1, 1, 1), rot=(1, 0, 0, 0), mat=-1, color=0):
    i = np.zeros(1, INSTANCE_DT)
    i["position"], i["rotation"], i["scale"] = pos, rot, scale
    i["matID"], i["objectID"], i["color"] = mat, 0, color
    return i


def _sun(direction, shadow=False):
    l = np.zeros(1, LIGHT_DT)
    l["type"], l["castShadow"], l["direction"] = 1, int(shadow), direction
    l["cutoff"], l["intensity"], l["active"] = -1, 1, 1
    return l


def _slab_depth(view, res, centre, half):
    """closed form: first hit of every primary ray with an axis-aligned box"""
    d = primary_rays(view[0], res)
    lo, hi = np.array(centre) - half, np.array(centre) + half
    with np.errstate(divide="ignore", invalid="ignore"):
        t0, t1 = lo / d, hi / d
    near = np.minimum(t0, t1).max(-1)
    far = np.maximum(t0, t1).min(-1)
    return np.where((near <= far) & (far > 0), np.maximum(near, 0), 0)


@pytest.mark.parametrize("res", [8, 33, 64])
def test_depth_of_a_box_in_closed_form(res):
    geo = cube_geometry()
    view = _view()
    scale = (1.5, 1.0, 0.5)
    inst = _instance([0.25, 6, -0.5], scale=scale)
    _, depth = ref_render(geo, 1, inst, [0], [1], view, _sun([0, 1, 0]), [0], [1], res)
    want = _slab_depth(view, res, [0.25, 6, -0.5], np.array(scale) / 2)
    hit = want > 0
    # (silhouette pixels may fall either way in fp32)
    agree = (depth[0] > 0) == hit
    assert agree.mean() > 0.995
    both = hit & (depth[0] > 0)
    assert both.sum() > 0
    assert np.allclose(depth[0][both], want[both], rtol=2e-6)


def test_colour_material_override_and_lambert():
    geo = cube_geometry()      # material 0 = (0.25, 0.5, 1.0)
    view = _view()
    res = 16
    centre = (res // 2, res // 2)
    # light along the view direction: the facing side is fully lit
    rgb, _ = ref_render(geo, 1, _instance([0, 4, 0]), [0], [1], view, _sun([0, 1, 0]),
                        [0], [1], res)
    assert tuple(rgb[0][centre]) == (63, 127, 255, 255)
    # 60 degrees off the normal: cos = 0.5
    s = np.sin(np.pi / 3)
    rgb, _ = ref_render(geo, 1, _instance([0, 4, 0]), [0], [1], view,
                        _sun([s, 0.5, 0]), [0], [1], res)
    assert np.abs(rgb[0][centre][:3].astype(int) - np.array([31, 63, 127])).max() <= 1
    # grazing light: the 0.2 ambient floor
    rgb, _ = ref_render(geo, 1, _instance([0, 4, 0]), [0], [1], view, _sun([1, 0, 0]),
                        [0], [1], res)
    assert tuple(rgb[0][centre][:3]) == (12, 25, 51)
    # colour override (-2): 0xRRGGBB
    rgb, _ = ref_render(geo, 1, _instance([0, 4, 0], mat=-2, color=0xFF204080), [0], [1],
                        view, _sun([0, 1, 0]), [0], [1], res)
    assert tuple(rgb[0][centre]) == (0x20, 0x40, 0x80, 255)
    # a miss is black with depth 0
    assert tuple(rgb[0][0, 0]) == (0, 0, 0, 255)


def test_shadow_ray_and_nearest_of_two():
    geo = cube_geometry()
    view = _view()
    res = 16
    centre = (res // 2, res // 2)
    rays = primary_rays(view[0], res)
    # a blocker between the eye and a big box
    inst = np.concatenate([_instance([0, 8, 0], scale=(4, 1, 4)),
                           _instance([0, 3, 0], scale=(0.6, 0.6, 0.6))])
    rgb, depth = ref_render(geo, 1, inst, [0], [2], view, _sun([0, 1, 0], shadow=True),
                            [0], [1], res)
    # rays through the image centre hit the blocker first
    assert abs(depth[0][centre] - 2.7 / rays[centre][1]) < 1e-5
    # the pixel next to it passes the blocker and sees the big box, lit: its
    # shadow ray towards -y misses the blocker
    beside = (res // 2, res // 2 + 1)
    assert abs(depth[0][beside] - 7.5 / rays[beside][1]) < 1e-5
    assert tuple(rgb[0][beside][:3]) == (63, 127, 255)
    # light placed so that the blocker sits between it and that point: ambient only
    p = rays[beside] * (7.5 / rays[beside][1])
    towards_light = np.array([0, 3, 0]) - p
    shadowed, _ = ref_render(geo, 1, inst, [0], [2], view,
                             _sun(list(-towards_light), shadow=True), [0], [1], res)
    assert tuple(shadowed[0][beside][:3]) == (12, 25, 51)
    # the same light without shadow casting: Lambert term only (the reference
    # does not normalise a directional light's direction, bvh_raycast.cpp:868)
    unit = towards_light / np.linalg.norm(towards_light)
    lit, _ = ref_render(geo, 1, inst, [0], [2], view, _sun(list(-unit), shadow=False),
                        [0], [1], res)
    want = np.floor(-unit[1] * np.array([0.25, 0.5, 1.0]) * 255)
    assert np.abs(lit[0][beside][:3].astype(int) - want).max() <= 1
    long_dir, _ = ref_render(geo, 1, inst, [0], [2], view,
                             _sun(list(-towards_light), shadow=False), [0], [1], res)
    assert tuple(long_dir[0][beside][:3]) == (63, 127, 255)
    # light from behind the big box: the seen face points away from it
    back, _ = ref_render(geo, 1, inst, [0], [2], view, _sun([0, -1, 0], shadow=True),
                         [0], [1], res)
    assert tuple(back[0][beside][:3]) == (12, 25, 51)


def test_two_worlds_do_not_see_each_other():
    geo = cube_geometry()
    views = np.concatenate([_view(), _view()])
    views["worldIDX"] = [0, 1]
    inst = np.concatenate([_instance([0, 3, 0]), _instance([0, 6, 0])])
    inst["worldIDX"] = [0, 1]
    lights = np.concatenate([_sun([0, 1, 0]), _sun([0, 1, 0])])
    res = 16
    _, depth = ref_render(geo, 2, inst, [0, 1], [1, 1], views, lights, [0, 1], [1, 1], res)
    c = (res // 2, res // 2)
    dy = primary_rays(views[0], res)[c][1]
    assert abs(depth[0][c] - 2.5 / dy) < 1e-5 and abs(depth[1][c] - 5.5 / dy) < 1e-5


def _quat_rot(q, v):
    w, p = q[0], q[1:]
    a = np.cross(p, v)
    return v + 2.0 * (a * w + np.cross(p, a))


def test_random_scene_of_every_mesh_kind_against_brute_force():
    """Rotated, scaled instances of the four render_prep meshes (box, 168-triangle
    ellipsoid, cylinder, wedge: bottom-level trees several levels deep in the
    shim's builder) against a float64 Moeller-Trumbore over all triangles with
    the reference's back-face culling."""
    import ctypes as C
    from madrona_amd.simlib import ref_lib_path
    from raycast_utils import Geometry
    lib_path = ref_lib_path("render_prep")
    if not os.path.exists(lib_path):
        pytest.skip("oracle/_ref/librender_prep_ref.so not built")
    lib = C.CDLL(lib_path)
    f = lib.sim_render_geometry
    f.restype = C.c_int32
    f.argtypes = [C.c_void_p] * 7
    counts = np.zeros(3, np.uint32)
    n_obj = f(None, None, None, None, None, None, counts.ctypes.data)
    verts = np.zeros((counts[0], 3), np.float32)
    idx = np.zeros((counts[1], 3), np.uint32)
    voff = np.zeros(n_obj + 1, np.uint32)
    toff = np.zeros(n_obj + 1, np.uint32)
    mats = np.zeros((counts[2], 3), np.float32)
    omat = np.zeros(n_obj, np.int32)
    f(verts.ctypes.data, idx.ctypes.data, voff.ctypes.data, toff.ctypes.data,
      mats.ctypes.data, omat.ctypes.data, None)
    geo = Geometry(verts, idx, voff, toff, omat, mats)

    rng = np.random.default_rng(5)
    n = 14
    inst = np.zeros(n, INSTANCE_DT)
    inst["position"] = np.stack([rng.uniform(-5, 5, n), rng.uniform(3, 12, n),
                                 rng.uniform(-3, 3, n)], -1)
    q = rng.normal(size=(n, 4))
    inst["rotation"] = q / np.linalg.norm(q, axis=1, keepdims=True)
    inst["scale"] = rng.uniform(0.5, 2.0, (n, 3))
    inst["matID"], inst["objectID"] = -1, np.arange(n) % n_obj
    view = _view()
    res = 48
    _, depth = ref_render(geo, 1, inst, [0], [n], view, _sun([0, 1, 0]), [0], [1], res)

    tris = []
    for i in inst:
        o = int(i["objectID"])
        v = geo.vertices[voff[o]:voff[o + 1]].astype(np.float64) * i["scale"]
        world = np.array([_quat_rot(i["rotation"].astype(np.float64), p) for p in v])
        tris.append((world + i["position"])[geo.indices[toff[o]:toff[o + 1]]])
    T = np.concatenate(tris)
    e1, e2 = T[:, 1] - T[:, 0], T[:, 2] - T[:, 0]
    normal = np.cross(e1, e2)
    rays = primary_rays(view[0], res)
    want = np.zeros((res, res))
    for py in range(res):
        for px in range(res):
            d = rays[py, px]
            pv = np.cross(d, e2)
            det = (e1 * pv).sum(-1)
            with np.errstate(all="ignore"):
                inv = 1.0 / det
                u = (-T[:, 0] * pv).sum(-1) * inv
                qv = np.cross(-T[:, 0], e1)
                v = (qv * d).sum(-1) * inv
                t = (e2 * qv).sum(-1) * inv
            ok = ((np.abs(det) > 1e-14) & (u >= 0) & (v >= 0) & (u + v <= 1) & (t > 0) &
                  ((normal * d).sum(-1) < 0))        # front faces only
            want[py, px] = t[ok].min() if ok.any() else 0.0
    hit = want > 0
    assert hit.mean() > 0.1
    agree = (depth[0] > 0) == hit
    assert agree.mean() > 0.995, float(agree.mean())
    both = hit & (depth[0] > 0)
    assert np.allclose(depth[0][both], want[both], rtol=1e-4)
