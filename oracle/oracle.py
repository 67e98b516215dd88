"""TEST INFRASTRUCTURE: ctypes binding of the CPU oracles (oracle/librt_oracle.so, liboctree_oracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; it is the
checker, never the thing measured or shipped.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_OLIB = None


def build(force: bool = False, targets=("librt_oracle.so", "liboctree_oracle.so")) -> None:
    targets = [t for t in targets if force or not os.path.exists(os.path.join(HERE, t))]
    if targets:
        r = subprocess.run(["make", "-C", HERE] + (["-B"] if force else []) + targets, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)


class _Scene(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("normals", C.c_void_p), ("tri_refs", C.c_void_p), ("nodes", C.c_void_p),
                ("n_vert", C.c_uint64), ("n_ref", C.c_uint64), ("n_words", C.c_uint64),
                ("bounds_min", C.c_float * 3), ("bounds_max", C.c_float * 3),
                ("sky", C.c_void_p), ("sky_w", C.c_int32), ("sky_h", C.c_int32)]


class _Frame(C.Structure):
    _fields_ = [("u", C.c_float * 70), ("width", C.c_int32), ("height", C.c_int32), ("row_begin", C.c_int32), ("row_end", C.c_int32),
                ("accum_index", C.c_uint32), ("blend", C.c_int32), ("ndc_x", C.c_void_p), ("ndc_y", C.c_void_p),
                ("rgba", C.c_void_p), ("depth", C.c_void_p), ("kind", C.c_void_p), ("slot", C.c_void_p), ("hit_t", C.c_void_p),
                ("nrays", C.c_void_p), ("seed_out", C.c_void_p), ("debug", C.c_void_p), ("counters", C.c_uint64 * 8), ("threads", C.c_int32)]


def _lib():
    global _LIB
    if _LIB is None:
        build(targets=("librt_oracle.so",))
        _LIB = C.CDLL(os.path.join(HERE, "librt_oracle.so"))
        _LIB.oracle_render.argtypes = [C.POINTER(_Scene), C.POINTER(_Frame)]
        _LIB.oracle_render.restype = C.c_int
        _LIB.oracle_math.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        _LIB.oracle_sky.argtypes = [C.POINTER(_Scene), C.c_void_p, C.c_void_p, C.c_uint64]
    return _LIB


COUNTER_NAMES = ["rays", "node_pops", "triangle_tests", "child_box_tests", "child_pointer_words", "triangle_hits_shaded", "rays_entering_octree"]


class OracleScene:
    """Holds contiguous copies of the scene arrays (same contract as rt_scene_create)."""

    def __init__(self, scene: dict, sky=None):
        self.v = np.ascontiguousarray(scene["vertices"], np.float32).reshape(-1, 4)
        self.n = np.ascontiguousarray(scene["normals"], np.float32).reshape(-1, 4)
        self.t = np.ascontiguousarray(scene["tri_refs"], np.uint32).reshape(-1, 4)
        self.k = np.ascontiguousarray(scene["nodes"], np.uint32).ravel()
        self.sky = None if sky is None else np.ascontiguousarray(sky, np.uint8)
        s = _Scene()
        s.vertices, s.normals, s.tri_refs, s.nodes = self.v.ctypes.data, self.n.ctypes.data, self.t.ctypes.data, self.k.ctypes.data
        s.n_vert, s.n_ref, s.n_words = len(self.v), len(self.t), len(self.k)
        s.bounds_min[:] = [float(x) for x in np.asarray(scene["bounds_min"], np.float32)]
        s.bounds_max[:] = [float(x) for x in np.asarray(scene["bounds_max"], np.float32)]
        if self.sky is not None:
            assert self.sky.ndim == 4 and self.sky.shape[0] == 6 and self.sky.shape[3] == 4
            s.sky, s.sky_h, s.sky_w = self.sky.ctypes.data, self.sky.shape[1], self.sky.shape[2]
        self.c = s

    def render(self, uniforms70, width, height, *, accum_index=1, blend=True, rgba=None, rows=None, ndc=None, taps=True, threads=None, debug=False):
        fr = _Frame()
        u = np.ascontiguousarray(uniforms70, np.float32)
        assert u.size == 70
        C.memmove(fr.u, u.ctypes.data, 280)
        fr.width, fr.height = width, height
        fr.row_begin, fr.row_end = (0, height) if rows is None else rows
        fr.accum_index, fr.blend = accum_index, int(blend)
        out = {}
        out["rgba"] = np.zeros((height, width, 4), np.float32) if rgba is None else np.ascontiguousarray(rgba, np.float32)
        out["depth"] = np.zeros((height, width), np.float32)
        fr.rgba, fr.depth = out["rgba"].ctypes.data, out["depth"].ctypes.data
        if ndc is not None:
            nx, ny = (np.ascontiguousarray(a, np.float32) for a in ndc)
            assert nx.shape == (height, width) and ny.shape == (height, width)
            fr.ndc_x, fr.ndc_y = nx.ctypes.data, ny.ctypes.data
        if taps:
            out["kind"] = np.zeros((height, width), np.uint32); fr.kind = out["kind"].ctypes.data
            out["slot"] = np.zeros((height, width), np.uint32); fr.slot = out["slot"].ctypes.data
            out["t"] = np.zeros((height, width), np.float32); fr.hit_t = out["t"].ctypes.data
            out["nrays"] = np.zeros((height, width), np.uint32); fr.nrays = out["nrays"].ctypes.data
            out["seed"] = np.zeros((height, width), np.float32); fr.seed_out = out["seed"].ctypes.data
        if debug:
            out["debug"] = np.zeros((height, width, 32), np.float32); fr.debug = out["debug"].ctypes.data
        fr.threads = threads if threads is not None else min(16, os.cpu_count() or 1)
        rc = _lib().oracle_render(C.byref(self.c), C.byref(fr))
        if rc != 0:
            raise RuntimeError(f"oracle_render failed rc={rc}")
        out["counters"] = {n: int(fr.counters[i]) for i, n in enumerate(COUNTER_NAMES)}
        return out

    def sample_sky(self, dirs):
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        o = np.zeros_like(d)
        _lib().oracle_sky(C.byref(self.c), d.ctypes.data, o.ctypes.data, len(d))
        return o


def depth_attachment(depth) -> np.ndarray:
    """Value the reference's GL_DEPTH_COMPONENT24 attachment (/root/reference/app/core/src/Framebuffer.cpp:10-12) stores
    for a gl_FragDepth plane, as llvmpipe converts it: clamp to [0, 1], multiply by 2^24 - 1 IN FLOAT (one rounding: below
    0.5 the product is rounded to a multiple of 1/2 or finer), then round to the nearest integer, ties to even.  Pinned by
    the attachment read-backs of the llvmpipe fixtures (tests/test_oracle_golden.py)."""
    d = np.asarray(depth, dtype=np.float32)
    d = np.where(np.isnan(d), np.float32(0.0), np.clip(d, np.float32(0.0), np.float32(1.0))).astype(np.float32)
    return np.rint(d * np.float32(16777215.0)).astype(np.uint32)


def overlay_composite(color, depth24, frag_color, frag_depth24, frag_mask=None):
    """Per-fragment operations of the reference's overlay draws (/root/reference/app/core/src/App.cpp:54,142-145):
    depth test GL_LEQUAL against the attachment, depth writes on, blending off.  Returns new (color, depth24)."""
    color, depth24 = np.array(color, dtype=np.float32), np.array(depth24, dtype=np.uint32)
    fz = np.asarray(frag_depth24, dtype=np.uint32)
    ok = fz <= depth24
    if frag_mask is not None:
        ok &= np.asarray(frag_mask) != 0
    color[ok] = np.asarray(frag_color, dtype=np.float32)[ok]
    depth24[ok] = fz[ok]
    return color, depth24


def math(op: str, x, y=None):
    ops = {"sin": 0, "cos": 1, "log2": 2, "exp2": 3, "pow": 4}
    x = np.ascontiguousarray(x, np.float32)
    yy = np.ascontiguousarray(y if y is not None else np.zeros_like(x), np.float32)
    o = np.zeros_like(x)
    _lib().oracle_math(ops[op], x.ctypes.data, yy.ctypes.data, o.ctypes.data, x.size)
    return o


# ---------------------------------------------------------------------------------------------
# Scene-build oracle (octree_oracle.cpp)
# ---------------------------------------------------------------------------------------------
class _OctResult(C.Structure):
    _fields_ = [("nodes", C.POINTER(C.c_uint32)), ("n_words", C.c_uint64), ("tri_refs", C.POINTER(C.c_uint32)), ("n_ref", C.c_uint64),
                ("bounds_min", C.c_float * 3), ("bounds_max", C.c_float * 3), ("max_depth", C.c_uint64)]


def _olib():
    global _OLIB
    if _OLIB is None:
        build(targets=("liboctree_oracle.so",))
        _OLIB = C.CDLL(os.path.join(HERE, "liboctree_oracle.so"))
        _OLIB.octree_oracle_build.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(_OctResult)]
        _OLIB.octree_oracle_build.restype = C.c_int
        _OLIB.octree_oracle_free.argtypes = [C.POINTER(_OctResult)]
        _OLIB.octree_oracle_tri_box.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _OLIB.octree_oracle_tri_box.restype = C.c_int
        _OLIB.mesh_oracle_build.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _OLIB.mesh_oracle_build.restype = C.c_int
    return _OLIB


def parse_obj(path: str):
    """Minimal OBJ reader for the oracle: 'v', optional 'vn', 'f' records (a, a/b, a//c, a/b/c forms) with three or four
    vertices.  The reference reads OBJ through tinyobjloader 2.0.0-rc10 with triangulation on (Mesh.cpp:12-14), a
    dependency that is not part of the reference checkout: its rule for quads -- cut along the shorter diagonal, squared
    lengths in float, (0,1,2)+(0,2,3) if |v0v2|^2 < |v1v3|^2 else (0,1,3)+(1,2,3) -- is restated from the published
    source and is PARITY UNPINNED (no fixture of the reference covers it)."""
    pos, nrm, faces = [], [], []
    with open(path) as f:
        for line in f:
            if line.startswith("v "):
                pos.append([float(t) for t in line.split()[1:4]])
            elif line.startswith("vn "):
                nrm.append([float(t) for t in line.split()[1:4]])
            elif line.startswith("f "):
                idx = [int(t.split("/")[0]) for t in line.split()[1:]]
                idx = [i - 1 if i > 0 else len(pos) + i for i in idx]
                if len(idx) == 3:
                    faces.append(idx)
                elif len(idx) == 4:
                    v = [np.asarray(pos[i], np.float64).astype(np.float32) for i in idx]
                    e02, e13 = v[2] - v[0], v[3] - v[1]
                    sqr02 = np.float32(np.float32(e02[0] * e02[0] + e02[1] * e02[1]) + e02[2] * e02[2])
                    sqr13 = np.float32(np.float32(e13[0] * e13[0] + e13[1] * e13[1]) + e13[2] * e13[2])
                    order = (0, 1, 2, 0, 2, 3) if sqr02 < sqr13 else (0, 1, 3, 1, 2, 3)
                    faces.append([idx[k] for k in order[:3]])
                    faces.append([idx[k] for k in order[3:]])
                else:
                    raise ValueError("faces with three or four vertices only")
    return (np.asarray(pos, np.float64).astype(np.float32), np.asarray(nrm, np.float64).astype(np.float32) if nrm else None,
            np.asarray(faces, np.uint32))


def mesh_build(pos, file_normals, faces, material_id: int, transform_cm):
    """Mesh.cpp:39-100 -> (vertices[n,4], normals[n,4], triangles[m,4])."""
    pos = np.ascontiguousarray(pos, np.float32); faces = np.ascontiguousarray(faces, np.uint32)
    fn = None if file_normals is None else np.ascontiguousarray(file_normals, np.float32)
    T = np.ascontiguousarray(transform_cm, np.float32).ravel()
    n, m = len(pos), len(faces)
    v = np.zeros((n, 4), np.float32); nn = np.zeros((n, 4), np.float32); t = np.zeros((m, 4), np.uint32)
    rc = _olib().mesh_oracle_build(pos.ctypes.data, None if fn is None else fn.ctypes.data, n, faces.ctypes.data, m, material_id, T.ctypes.data,
                                   v.ctypes.data, nn.ctypes.data, t.ctypes.data)
    if rc:
        raise RuntimeError(f"mesh_oracle_build rc={rc}")
    return v, nn, t


def octree_build(vertices, triangles, depth_limit=18, max_per_node=10, max_per_leaf=6, keep_after=6) -> dict:
    """Octree.cpp:203-256 -> dict(nodes, tri_refs, bounds_min, bounds_max, max_depth).  Defaults: App.cpp:22."""
    v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 4)
    t = np.ascontiguousarray(triangles, np.uint32).reshape(-1, 4)
    r = _OctResult()
    rc = _olib().octree_oracle_build(v.ctypes.data, len(v), t.ctypes.data, len(t), depth_limit, max_per_node, max_per_leaf, keep_after, C.byref(r))
    if rc:
        raise RuntimeError(f"octree_oracle_build rc={rc}")
    out = dict(nodes=np.ctypeslib.as_array(r.nodes, (r.n_words,)).copy(), tri_refs=np.ctypeslib.as_array(r.tri_refs, (r.n_ref * 4,)).copy().reshape(-1, 4),
               bounds_min=np.array(r.bounds_min[:], np.float32), bounds_max=np.array(r.bounds_max[:], np.float32), max_depth=int(r.max_depth))
    _olib().octree_oracle_free(C.byref(r))
    return out


def tri_box(center, halfsize, tri) -> bool:
    c = np.ascontiguousarray(center, np.float32); h = np.ascontiguousarray(halfsize, np.float32); t = np.ascontiguousarray(tri, np.float32).ravel()
    return bool(_olib().octree_oracle_tri_box(c.ctypes.data, h.ctypes.data, t.ctypes.data))


def present_rgba8(color: np.ndarray) -> np.ndarray:
    """The window's 8-bit image of an RGBA32F attachment (Framebuffer.cpp:33-35 on llvmpipe): rint(float32(clamp(c, 0, 1) * 255)), NaN -> 0."""
    c = np.nan_to_num(np.clip(np.asarray(color, np.float32), 0, 1), nan=0.0).astype(np.float32)
    return np.rint((c * np.float32(255.0)).astype(np.float32)).astype(np.uint8)
