"""CPU: what the executor builds from a geometry description (host code of the
batch ray caster): bottom-level BVH sizes, object bounds, and the
detection of objects that are their own axis-aligned bounds (they take the slab
path of the trace kernel).  No GPU involved: mwhip_render_geometry_info."""
import ctypes as C

import numpy as np
import pytest

from madrona_amd.simlib import runtime_lib
from raycast_utils import cube_geometry


class RenderGeometry(C.Structure):
    _fields_ = [("num_objects", C.c_uint32), ("num_materials", C.c_uint32),
                ("vertices", C.c_void_p), ("indices", C.c_void_p),
                ("object_vertex_offset", C.c_void_p),
                ("object_triangle_offset", C.c_void_p),
                ("object_material", C.c_void_p), ("material_color", C.c_void_p),
                # per-triangle materials and textures (optional: NULL / 0)
                ("vertex_uv", C.c_void_p), ("triangle_material", C.c_void_p),
                ("material_texture", C.c_void_p), ("num_textures", C.c_uint32),
                ("pad_", C.c_uint32), ("textures", C.c_void_p)]


def _info(objects):
    """objects: list of (vertices [V,3], triangles [T,3])"""
    verts = np.concatenate([np.asarray(v, np.float32).reshape(-1, 3) for v, _ in objects])
    tris = np.concatenate([np.asarray(t, np.uint32).reshape(-1, 3) for _, t in objects])
    voff = np.cumsum([0] + [len(np.asarray(v).reshape(-1, 3)) for v, _ in objects]).astype(np.uint32)
    toff = np.cumsum([0] + [len(np.asarray(t).reshape(-1, 3)) for _, t in objects]).astype(np.uint32)
    g = RenderGeometry(len(objects), 0, verts.ctypes.data, tris.ctypes.data,
                       voff.ctypes.data, toff.ctypes.data, None, None,
                       None, None, None, 0, 0, None)
    rt = runtime_lib()
    rt.mwhip_render_geometry_info.restype = C.c_int
    rt.mwhip_render_geometry_info.argtypes = [C.POINTER(RenderGeometry), C.c_void_p,
                                              C.c_void_p, C.c_void_p]
    nodes = np.zeros(len(objects), np.uint32)
    is_box = np.zeros(len(objects), np.uint32)
    bounds = np.zeros((len(objects), 6), np.float32)
    rc = rt.mwhip_render_geometry_info(C.byref(g), nodes.ctypes.data, is_box.ctypes.data,
                                       bounds.ctypes.data)
    return rc, nodes, is_box, bounds


def _cube(lo=(-0.5, -0.5, -0.5), hi=(0.5, 0.5, 0.5)):
    g = cube_geometry()
    v = g.vertices * (np.array(hi) - np.array(lo)) + (np.array(hi) + np.array(lo)) / 2
    return v, g.indices


def _sphere(n=8):
    v = [[0, 0, -1]] + [[np.sin(p) * np.cos(t), np.sin(p) * np.sin(t), -np.cos(p)]
                        for p in np.pi * np.arange(1, n) / n
                        for t in 2 * np.pi * np.arange(n) / n] + [[0, 0, 1]]
    ring = lambda s, k: 1 + (s - 1) * n + k % n
    t = []
    for k in range(n):
        t += [(0, ring(1, k + 1), ring(1, k)), (len(v) - 1, ring(n - 1, k), ring(n - 1, k + 1))]
        for s in range(1, n - 1):
            t += [(ring(s, k), ring(s, k + 1), ring(s + 1, k + 1)),
                  (ring(s, k), ring(s + 1, k + 1), ring(s + 1, k))]
    return v, t


def test_boxes_are_recognised_and_nothing_else_is():
    cube_v, cube_t = _cube()
    slab_v, slab_t = _cube((-2, 0, 1), (3, 0.25, 1.5))
    inside_out = cube_t[:, ::-1]                        # faces point inwards
    rot = np.array([[np.cos(.3), -np.sin(.3), 0], [np.sin(.3), np.cos(.3), 0], [0, 0, 1]])
    missing_face = cube_t[:10]                          # open box
    sphere_v, sphere_t = _sphere()
    rc, nodes, is_box, bounds = _info([
        (cube_v, cube_t), (slab_v, slab_t), (cube_v, inside_out), (cube_v @ rot.T, cube_t),
        (cube_v, missing_face), (sphere_v, sphere_t)])
    assert rc == 0
    assert is_box.tolist() == [1, 1, 0, 0, 0, 0]
    assert np.allclose(bounds[0], [-.5, -.5, -.5, .5, .5, .5])
    assert np.allclose(bounds[1], [-2, 0, 1, 3, 0.25, 1.5])
    assert np.allclose(bounds[5], [-1, -1, -1, 1, 1, 1], atol=1e-6)
    # <= 4 triangles per leaf, two children per node: 12 triangles need >= 2
    # nodes, the 112-triangle sphere a tree of >= 14 leaves
    assert 2 <= nodes[0] <= 5 and nodes[5] >= 13


def test_malformed_geometry_is_refused():
    cube_v, cube_t = _cube()
    bad = cube_t.copy()
    bad[3, 1] = 99                                      # vertex that does not exist
    rc, *_ = _info([(cube_v, bad)])
    assert rc != 0
    assert b"vertex" in runtime_lib().mwhip_last_error()
