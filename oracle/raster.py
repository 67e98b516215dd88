"""TEST INFRASTRUCTURE: Python face of oracle/raster_oracle.c (the raster pass that follows the ray-tracing draw,
App.cpp:142-166), plus the inputs of that pass: the mesh as the reference's loader hands it to GL and the instance matrices."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(HERE, "libraster_oracle.so")
        if not os.path.exists(so):
            subprocess.run(["make", "-C", HERE, "libraster_oracle.so"], check=True, capture_output=True)
        _lib = C.CDLL(so)
        _lib.raster_oracle_draw.restype = C.c_int
        _lib.raster_oracle_draw.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    return _lib


def overlay_words(view_proj, worlds, world_its, light_dir, la, ld) -> np.ndarray:
    """viewProj[16] (column-major), light_dir[3], La[3], Ld[3], one pad word, then world[16] + worldIT[16] per draw (App.cpp:152-163)."""
    head = np.concatenate([np.asarray(view_proj, np.float32).reshape(16), np.asarray(light_dir, np.float32), np.asarray(la, np.float32), np.asarray(ld, np.float32), np.zeros(1, np.float32)])
    inst = [np.concatenate([np.asarray(w, np.float32).reshape(16), np.asarray(wi, np.float32).reshape(16)]) for w, wi in zip(worlds, world_its)]
    return np.concatenate([head] + inst).astype(np.float32)


def reference_instances():
    """The nine draws of App.cpp:157-164: world = translate(4 i, 4 + 4 j, 0) for i, j in -1..1 (i outer), worldIT = transpose(inverse(world)).
    Column-major 16-float arrays.  The inverse of a translation is the opposite translation, exactly; transposed it sits in the last row."""
    worlds, its = [], []
    for i in (-1, 0, 1):
        for j in (-1, 0, 1):
            t = np.array([4.0 * i, 4.0 + 4.0 * j, 0.0], np.float32)
            w = np.eye(4, dtype=np.float32); w[3, :3] = t              # w[c] = column c
            it = np.eye(4, dtype=np.float32); it[:3, 3] = -t           # row 3 of the matrix = element 3 of the columns
            worlds.append(w.reshape(16)); its.append(it.reshape(16))
    return worlds, its


REFERENCE_LIGHT = dict(light_dir=(-1.0, -2.0, -0.5), la=(0.0, 0.0, 0.0), ld=(1.0, 1.0, 1.0))      # App.hpp:79-81


from raytracer_amd.objmesh import load_obj  # noqa: E402,F401  (the loader is host-side product code; fixtures are generated through it)


def draw(vertices, indices, words, texture, color, depth24, *, want_ids=False):
    """Runs the pass over `color` [h, w, 4] f32 and `depth24` [h, w] u32 (row 0 = bottom) in place."""
    vertices = np.ascontiguousarray(vertices, np.float32); indices = np.ascontiguousarray(indices, np.uint32)
    words = np.ascontiguousarray(words, np.float32); texture = np.ascontiguousarray(texture, np.uint8)
    assert color.dtype == np.float32 and color.flags.c_contiguous and depth24.dtype == np.uint32 and depth24.flags.c_contiguous
    h, w = depth24.shape
    n_draws = (len(words) - 26) // 32
    ids = np.full((h, w), 0xFFFFFFFF, np.uint32) if want_ids else None
    rc = lib().raster_oracle_draw(vertices.ctypes.data, len(vertices), indices.ctypes.data, len(indices), words.ctypes.data, n_draws,
                                  texture.ctypes.data, texture.shape[1], texture.shape[0], color.ctypes.data, depth24.ctypes.data,
                                  ids.ctypes.data if want_ids else None, w, h)
    if rc:
        raise RuntimeError("raster_oracle_draw failed: %d" % rc)
    return ids
