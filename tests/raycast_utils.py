"""Test-side helpers for the batch ray caster: ctypes binding of
the reference's ray caster compiled for the host (oracle/_ref/libraycast_ref.so,
oracle/ref_shims/raycast_ref_shim.cpp) and small scene builders.  Test
infrastructure only."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libraycast_ref.so")

INSTANCE_DT = np.dtype([("position", "<f4", 3), ("rotation", "<f4", 4), ("scale", "<f4", 3),
                        ("matID", "<i4"), ("objectID", "<i4"), ("worldIDX", "<i4"),
                        ("color", "<u4"), ("pad", "<u4", 2)])
VIEW_DT = np.dtype([("position", "<f4", 3), ("rotation", "<f4", 4), ("xScale", "<f4"),
                    ("yScale", "<f4"), ("zNear", "<f4"), ("worldIDX", "<i4"), ("pad", "<u4")])
LIGHT_DT = np.dtype([("type", "u1"), ("castShadow", "u1"), ("pad0", "u1", 2),
                     ("position", "<f4", 3), ("direction", "<f4", 3), ("cutoff", "<f4"),
                     ("intensity", "<f4"), ("active", "u1"), ("pad1", "u1", 3)])
assert INSTANCE_DT.itemsize == 64 and VIEW_DT.itemsize == 48 and LIGHT_DT.itemsize == 40


class Geometry:
    """vertices [V,3] f32, indices [T,3] u32 (object-local), per-object offsets,
    object_materials [O] i32 (-1: none), material_colors [M,3] f32"""

    def __init__(self, vertices, indices, vertex_offsets, triangle_offsets,
                 object_materials, material_colors, vertex_uvs=None,
                 triangle_materials=None, material_textures=None, textures=()):
        self.vertices = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
        self.indices = np.ascontiguousarray(indices, np.uint32).reshape(-1, 3)
        self.vertex_offsets = np.ascontiguousarray(vertex_offsets, np.uint32)
        self.triangle_offsets = np.ascontiguousarray(triangle_offsets, np.uint32)
        self.object_materials = np.ascontiguousarray(object_materials, np.int32)
        self.material_colors = np.ascontiguousarray(material_colors, np.float32).reshape(-1, 3)
        self.num_objects = len(self.triangle_offsets) - 1
        # per-vertex uvs [V,2], per-triangle materials [T] (objects whose material
        # is -1), texture id per material [M], textures = [(width, height, rgba8
        # bytes [h,w,4])]
        self.vertex_uvs = None if vertex_uvs is None else \
            np.ascontiguousarray(vertex_uvs, np.float32).reshape(-1, 2)
        self.triangle_materials = None if triangle_materials is None else \
            np.ascontiguousarray(triangle_materials, np.int32)
        self.material_textures = None if material_textures is None else \
            np.ascontiguousarray(material_textures, np.int32)
        self.textures = list(textures)


def cube_geometry():
    v = np.array([[x, y, z] for z in (-.5, .5) for y in (-.5, .5) for x in (-.5, .5)], np.float32)
    quads = [(0, 2, 3, 1), (4, 5, 7, 6), (0, 1, 5, 4), (2, 6, 7, 3), (0, 4, 6, 2), (1, 3, 7, 5)]
    tris = [t for a, b, c, d in quads for t in ((a, b, c), (a, c, d))]
    return Geometry(v, tris, [0, 8], [0, 12], [0], [[0.25, 0.5, 1.0]])


def ref_render(geo, num_worlds, instances, inst_offsets, inst_counts, views, lights,
               light_offsets, light_counts, resolution, rgbd=True, threads=8):
    """The reference's bvhRaycastEntry over every pixel of every view.
    Returns (rgb [V,res,res,4] u8, depth [V,res,res] f32)."""
    lib = C.CDLL(REF_LIB)
    lib.raycast_ref_render.restype = C.c_int
    p = C.c_void_p
    lib.raycast_ref_render.argtypes = [C.c_uint32, p, p, p, p, p, C.c_uint32, p,
                                       C.c_uint32, p, p, p, p, C.c_uint32, p, p, p,
                                       C.c_uint32, C.c_uint32, C.c_uint32, p, p,
                                       p, p, p, C.c_uint32, p, p]
    instances = np.ascontiguousarray(instances)
    views = np.ascontiguousarray(views)
    lights = np.ascontiguousarray(lights)
    assert instances.dtype.itemsize == 64 or instances.shape[-1] == 64
    io = np.ascontiguousarray(inst_offsets, np.int32)
    ic = np.ascontiguousarray(inst_counts, np.int32)
    lo = np.ascontiguousarray(light_offsets, np.int32)
    lc = np.ascontiguousarray(light_counts, np.int32)
    nv = views.shape[0]
    rgb = np.zeros((nv, resolution, resolution, 4), np.uint8)
    depth = np.zeros((nv, resolution, resolution), np.float32)
    if lights.size == 0:
        lights = np.zeros(1, LIGHT_DT)
    tex_dims = np.array([[w, h] for w, h, _ in geo.textures] or [[0, 0]], np.uint32)
    texels = np.concatenate([np.frombuffer(bytes(px), np.uint8) for _, _, px in geo.textures]
                            or [np.zeros(4, np.uint8)])
    rc = lib.raycast_ref_render(
        geo.num_objects, geo.vertices.ctypes.data, geo.indices.ctypes.data,
        geo.vertex_offsets.ctypes.data, geo.triangle_offsets.ctypes.data,
        geo.object_materials.ctypes.data, len(geo.material_colors),
        geo.material_colors.ctypes.data, num_worlds, instances.ctypes.data,
        io.ctypes.data, ic.ctypes.data, views.ctypes.data, nv, lights.ctypes.data,
        lo.ctypes.data, lc.ctypes.data, resolution, 1 if rgbd else 0, threads,
        rgb.ctypes.data, depth.ctypes.data,
        None if geo.vertex_uvs is None else geo.vertex_uvs.ctypes.data,
        None if geo.triangle_materials is None else geo.triangle_materials.ctypes.data,
        None if geo.material_textures is None else geo.material_textures.ctypes.data,
        len(geo.textures), tex_dims.ctypes.data, texels.ctypes.data)
    assert rc == 0, f"raycast_ref_render failed: {rc}"
    return rgb, depth


def primary_rays(view, resolution):
    """numpy restatement of calculateOutRay (reference bvh_raycast.cpp:58-88) for
    an identity view rotation: directions [res,res,3] (row = pixel y)."""
    h = 1.0 / (-float(view["yScale"]))
    viewport = 2.0 * h
    forward = np.array([0, 1, 0], np.float64)
    u = np.array([1, 0, 0], np.float64)
    v = np.cross(forward, u)
    v /= np.linalg.norm(v)
    px = (np.arange(resolution) + 0.5) / resolution
    lower_left = -u * viewport / 2 - v * viewport / 2 + forward
    d = (lower_left[None, None, :] + px[None, :, None] * (u * viewport)[None, None, :] +
         px[:, None, None] * (v * viewport)[None, None, :])
    return d / np.linalg.norm(d, axis=-1, keepdims=True)
