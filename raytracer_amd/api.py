"""ctypes binding of librt_hotpath.so (include/rt_hotpath.h) and a thin host-side mirror of the
reference's scene interface.

Names follow the reference where it has them: ``Mesh(filename, material_id, transform)`` with
``m_vertecies / m_normals / m_triangles`` (/root/reference/app/core/include/Mesh.hpp:11-17) and
``Octree(meshes, depth_limit, max_triangles_per_node, max_triangles_per_leaf, keep_after)`` with
``m_vertecies / m_normals / m_compressed_node_buffer / m_compressed_triangles / GetMinBounds() /
GetMaxBounds()`` (include/Octree.hpp:27-36).  ``RenderPass`` plays the part of App::Render's
ray-tracing draw (src/App.cpp:96-140) including the progressive-accumulation counter
(App.cpp:29,90-92,104-105).

There is no fallback: if the shared library is missing or no HIP device is usable, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .params import Uniforms

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_HOTPATH_LIB", os.path.join(_HERE, "librt_hotpath.so"))   # override: development builds

RT_OK, RT_EINVAL, RT_ESCENE, RT_ENODEVICE, RT_ENOMEM, RT_ELIMIT = range(6)
RT_MEM_HOST, RT_MEM_DEVICE = 0, 1
HIT_MISS, HIT_SPHERE, HIT_TRIANGLE, HIT_CYLINDER, HIT_PORTAL_1, HIT_PORTAL_2 = range(6)

EXPORTED_SYMBOLS = [
    "rt_last_error", "rt_build_id", "rt_device_count", "rt_set_device", "rt_scene_create", "rt_skybox_set", "rt_render", "rt_render_sequence", "rt_render_noalloc", "rt_band_row_count", "rt_scene_stats",
    "rt_scene_set_counting", "rt_scene_set_kernel_timing", "rt_scene_kernel_time", "rt_scene_set_pipeline", "rt_scene_pipeline_state", "rt_scene_set_subframes", "rt_scene_set_frames_in_flight", "rt_scene_set_frames_per_launch", "rt_scene_set_sequence_batch", "rt_scene_set_traversal_order", "rt_scene_selection_state", "rt_scene_last_mode", "rt_scene_destroy", "rt_mesh_load_obj", "rt_mesh_from_arrays", "rt_mesh_free", "rt_octree_build", "rt_octree_build_device", "rt_octree_free", "rt_scene_create_from_meshes",
    # include/rt_camera.h (bound in glm_camera.py)
    "rt_vec3_normalize", "rt_camera_init", "rt_camera_set_view", "rt_camera_set_proj", "rt_camera_view_proj", "rt_camera_inv_view_proj", "rt_portal_difference_matrix", "rt_portal_ray",
    "rt_manipulator_init", "rt_manipulator_set_camera", "rt_manipulator_update", "rt_manipulator_key", "rt_manipulator_mouse_move", "rt_manipulator_mouse_wheel",
    "rt_uniforms_set_camera", "rt_uniforms_set_portals",
    # include/rt_overlay.h
    "rt_depth_to_attachment", "rt_overlay_composite", "rt_overlay_mesh_create", "rt_overlay_mesh_destroy", "rt_overlay_draw", "rt_overlay_draw_band", "rt_present_rgba8",
    # include/rt_dist.h (bound in tests/test_gpu_dist_cabi.py)
    "rt_dist_make_id", "rt_dist_init", "rt_dist_rank", "rt_dist_world_size", "rt_dist_destroy", "rt_dist_scene_create", "rt_dist_skybox_set",
    "rt_dist_frame_bands", "rt_dist_gather_frame",
]


class RtError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"rt_hotpath error {code}: {msg}")
        self.code = code


class rt_uniforms(C.Structure):
    _fields_ = [("words", C.c_float * 70)]


class rt_frame(C.Structure):
    _fields_ = [("u", rt_uniforms), ("width", C.c_uint32), ("height", C.c_uint32), ("accum_index", C.c_uint32), ("blend", C.c_uint32),
                ("band_rows", C.c_uint32), ("band_parts", C.c_uint32), ("band_part", C.c_uint32), ("compact_rows", C.c_uint32),
                ("ndc_x", C.c_void_p), ("ndc_y", C.c_void_p), ("stream", C.c_void_p)]


class rt_target(C.Structure):
    _fields_ = [("memory", C.c_int32), ("sync", C.c_int32), ("color", C.c_void_p), ("depth", C.c_void_p), ("hit_kind", C.c_void_p),
                ("hit_slot", C.c_void_p), ("hit_t", C.c_void_p), ("n_rays", C.c_void_p)]


class rt_stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("node_pops", C.c_uint64), ("triangle_tests", C.c_uint64), ("child_box_tests", C.c_uint64),
                ("triangle_hits_shaded", C.c_uint64), ("rays_entering_octree", C.c_uint64), ("sphere_tests", C.c_uint64),
                ("node_pops_from_lds", C.c_uint64), ("exact_fallbacks", C.c_uint64), ("exact_candidate_checks", C.c_uint64)]


class rt_mesh(C.Structure):
    _fields_ = [("vertices", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)), ("triangles", C.POINTER(C.c_uint32)),
                ("n_vert", C.c_size_t), ("n_tri", C.c_size_t)]


class rt_octree(C.Structure):
    _fields_ = [("vertices", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)), ("tri_refs", C.POINTER(C.c_uint32)), ("nodes", C.POINTER(C.c_uint32)),
                ("n_vert", C.c_size_t), ("n_ref", C.c_size_t), ("n_words", C.c_size_t), ("bounds_min", C.c_float * 3), ("bounds_max", C.c_float * 3),
                ("max_depth", C.c_size_t)]


PIPELINE_NAMES = ["megakernel", "wavefront", "persistent", "auto"]
ORDER_NAMES = ["reference", "near_first", "near_first_exact", "auto"]
CANDIDATE_NAMES = ["megakernel", "wavefront", "persistent", "wavefront+near_first_exact", "persistent+near_first_exact"]

_lib = None


def lib():
    """Load librt_hotpath.so; raises if it has not been built (see __graft_entry__.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RtError(RT_ENODEVICE, f"{LIB_PATH} is missing: build it with `make -C raytracer_amd/csrc` (there is no CPU fallback)")
        # PyTorch-ROCm wheels carry their own copy of the HIP runtime under another file name.  Loaded after this library's
        # it becomes a SECOND runtime in the process and finds no GPU; loaded first, the dynamic loader gives this library
        # the same copy (equal soname).  So a process that can use torch at all gets it loaded here, before the library.
        import importlib.util
        import sys
        if "torch" not in sys.modules and importlib.util.find_spec("torch") is not None:
            import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        L.rt_last_error.restype = C.c_char_p
        L.rt_build_id.restype = C.c_char_p
        L.rt_device_count.argtypes = [C.POINTER(C.c_int)]
        L.rt_set_device.argtypes = [C.c_int]
        L.rt_scene_create.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        L.rt_skybox_set.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int]
        L.rt_render.argtypes = [C.c_void_p, C.POINTER(rt_frame), C.POINTER(rt_target)]
        L.rt_render_noalloc.argtypes = [C.c_void_p, C.POINTER(rt_frame), C.POINTER(rt_target)]
        L.rt_render_sequence.argtypes = [C.c_void_p, C.POINTER(rt_frame), C.c_size_t, C.POINTER(rt_target)]
        L.rt_band_row_count.argtypes = [C.POINTER(rt_frame)]
        L.rt_band_row_count.restype = C.c_uint32
        L.rt_scene_stats.argtypes = [C.c_void_p, C.POINTER(rt_stats)]
        L.rt_scene_set_counting.argtypes = [C.c_void_p, C.c_int]
        L.rt_scene_set_kernel_timing.argtypes = [C.c_void_p, C.c_int]
        L.rt_scene_kernel_time.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
        L.rt_scene_set_pipeline.argtypes = [C.c_void_p, C.c_int]
        L.rt_scene_pipeline_state.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.rt_scene_set_subframes.argtypes = [C.c_void_p, C.c_int]
        L.rt_scene_set_frames_in_flight.argtypes = [C.c_void_p, C.c_int]
        L.rt_scene_set_frames_per_launch.argtypes = [C.c_void_p, C.c_int]
        L.rt_scene_set_sequence_batch.argtypes = [C.c_void_p, C.c_int]
        L.rt_scene_set_traversal_order.argtypes = [C.c_void_p, C.c_int]
        L.rt_scene_selection_state.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.rt_scene_last_mode.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.rt_scene_destroy.argtypes = [C.c_void_p]
        L.rt_mesh_load_obj.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.POINTER(rt_mesh)]
        L.rt_mesh_from_arrays.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.POINTER(rt_mesh)]
        L.rt_mesh_free.argtypes = [C.POINTER(rt_mesh)]
        L.rt_mesh_free.restype = None
        L.rt_octree_build.argtypes = [C.POINTER(rt_mesh), C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(rt_octree)]
        L.rt_octree_build_device.argtypes = [C.POINTER(rt_mesh), C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.POINTER(rt_octree)]
        L.rt_octree_free.argtypes = [C.POINTER(rt_octree)]
        L.rt_scene_create_from_meshes.argtypes = [C.POINTER(rt_mesh), C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        L.rt_octree_free.restype = None
        _lib = L
    return _lib


def _check(rc: int):
    if rc != RT_OK:
        raise RtError(rc, lib().rt_last_error().decode(errors="replace"))


def build_id() -> str:
    """rt_build_id: digest of the kernel sources the loaded library was built from."""
    return lib().rt_build_id().decode()


def source_id() -> str:
    """The same digest computed from the source files beside this package (equal to build_id() when the library is up to date)."""
    import hashlib
    h = hashlib.sha256()
    for name in ("rt_kernels.hip", "rt_internal.h", "rt_api.cpp"):
        with open(os.path.join(_HERE, "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def device_count() -> int:
    n = C.c_int(0)
    rc = lib().rt_device_count(C.byref(n))
    return n.value if rc == RT_OK else 0


def _is_tensor(x) -> bool:
    return hasattr(x, "data_ptr") and hasattr(x, "is_cuda")


def _plane_ptr(x, dtype, n, name):
    """Pointer, memory space and a keep-alive reference of a plane given as a numpy array or a contiguous GPU tensor."""
    if _is_tensor(x):
        if not x.is_cuda or not x.is_contiguous() or x.numel() != n:
            raise ValueError(f"{name}: a contiguous GPU tensor of {n} elements is required")
        return x.data_ptr(), RT_MEM_DEVICE, x
    a = np.ascontiguousarray(x, dtype=dtype)
    if a.size != n:
        raise ValueError(f"{name}: {n} elements are required")
    return a.ctypes.data, RT_MEM_HOST, a


def depth_to_attachment(depth, out=None, stream: int | None = None):
    """rt_depth_to_attachment: the 24-bit values the reference's depth attachment holds for a gl_FragDepth plane
    (Framebuffer.cpp:10-12).  numpy in -> numpy uint32 out (staged); GPU tensor in -> GPU int32 tensor out (same bits)."""
    L = lib()
    L.rt_depth_to_attachment.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    if _is_tensor(depth):
        import torch
        if depth.dtype != torch.float32:
            raise ValueError("depth: a float32 tensor is required")
        if out is None:
            out = torch.empty(depth.shape, dtype=torch.int32, device=depth.device)
        n = depth.numel()
        p_in, mem, _k1 = _plane_ptr(depth, None, n, "depth")
        p_out, _, _k2 = _plane_ptr(out, None, n, "out")
        _check(L.rt_depth_to_attachment(p_in, p_out, n, mem, stream if stream is not None else torch.cuda.current_stream().cuda_stream))
        return out
    d = np.ascontiguousarray(depth, dtype=np.float32)
    z = np.empty(d.shape, dtype=np.uint32) if out is None else out
    _check(L.rt_depth_to_attachment(d.ctypes.data, z.ctypes.data, d.size, RT_MEM_HOST, None))
    return z


def present_rgba8(color, out=None, stream: int | None = None):
    """rt_present_rgba8: the 8-bit image the window receives from the RGBA32F attachment (Framebuffer.cpp:33-35).  numpy [..., 4]
    float32 in -> numpy uint8 out; GPU tensor in -> GPU uint8 tensor out."""
    L = lib()
    L.rt_present_rgba8.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    if _is_tensor(color):
        import torch
        if color.dtype != torch.float32 or color.shape[-1] != 4:
            raise ValueError("color: a float32 tensor [..., 4] is required")
        if out is None:
            out = torch.empty(color.shape, dtype=torch.uint8, device=color.device)
        n = color.numel() // 4
        p_in, mem, _k1 = _plane_ptr(color, None, 4 * n, "color")
        p_out, _, _k2 = _plane_ptr(out, None, 4 * n, "out")
        _check(L.rt_present_rgba8(p_in, p_out, n, mem, stream if stream is not None else torch.cuda.current_stream().cuda_stream))
        return out
    c = np.ascontiguousarray(color, dtype=np.float32)
    o = np.empty(c.shape, dtype=np.uint8) if out is None else out
    _check(L.rt_present_rgba8(c.ctypes.data, o.ctypes.data, c.size // 4, RT_MEM_HOST, None))
    return o


def overlay_composite(color, depth24, frag_color, frag_depth24, frag_mask=None, stream: int | None = None):
    """rt_overlay_composite: GL_LEQUAL depth-tested write of one layer of rasterised fragments into `color` ([..., 4]
    float32) and `depth24`, in place (App.cpp:54,142-145).  All planes numpy arrays, or all contiguous GPU tensors
    (depth planes int32 holding the 24-bit values, mask uint8)."""
    L = lib()
    L.rt_overlay_composite.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    tensors = _is_tensor(color)
    n = depth24.numel() if tensors else int(np.asarray(depth24).size)
    if tensors:
        import torch
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
    elif not (isinstance(color, np.ndarray) and color.dtype == np.float32 and color.flags.c_contiguous and isinstance(depth24, np.ndarray) and depth24.dtype == np.uint32 and depth24.flags.c_contiguous):
        raise ValueError("color must be a C-contiguous float32 array and depth24 a C-contiguous uint32 array (they are updated in place)")
    pc, mem, _k1 = _plane_ptr(color, np.float32, 4 * n, "color")
    pz, _, _k2 = _plane_ptr(depth24, np.uint32, n, "depth24")
    pfc, m3, _k3 = _plane_ptr(frag_color, np.float32, 4 * n, "frag_color")
    pfz, m4, _k4 = _plane_ptr(frag_depth24, np.uint32, n, "frag_depth24")
    pm, m5, _k5 = (None, mem, None) if frag_mask is None else _plane_ptr(frag_mask, np.uint8, n, "frag_mask")
    if len({mem, m3, m4, m5}) != 1:
        raise ValueError("all planes must live in the same memory space")
    _check(L.rt_overlay_composite(pc, pz, pfc, pfz, pm, n, mem, stream))
    return color, depth24


class rt_overlay_pass(C.Structure):
    _fields_ = [("view_proj", C.c_float * 16), ("light_dir", C.c_float * 3), ("la", C.c_float * 3), ("ld", C.c_float * 3),
                ("n_draws", C.c_uint32), ("world", C.c_void_p), ("world_it", C.c_void_p)]


class OverlayMesh:
    """rt_overlay_mesh: the mesh and texture of the reference's raster pass (App.cpp:65-76: m_Suzanne and m_metalTextureID) on a
    HIP device.  vertices [n, 8] float32 (position, normal, texcoord), indices uint32 (triangles), texture [h, w, 4] uint8 with
    row 0 = the bottom row (GLUtils.cpp:187-202)."""

    def __init__(self, vertices, indices, texture, device: int = 0):
        L = lib()
        L.rt_overlay_mesh_create.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]
        L.rt_overlay_mesh_destroy.argtypes = [C.c_void_p]
        L.rt_overlay_draw.argtypes = [C.c_void_p, C.POINTER(rt_overlay_pass), C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
        L.rt_overlay_draw_band.argtypes = [C.c_void_p, C.POINTER(rt_overlay_pass), C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
        v = np.ascontiguousarray(vertices, np.float32); i = np.ascontiguousarray(indices, np.uint32).reshape(-1); t = np.ascontiguousarray(texture, np.uint8)
        if v.ndim != 2 or v.shape[1] != 8 or t.ndim != 3 or t.shape[2] != 4:
            raise ValueError("vertices must be [n, 8] and the texture [h, w, 4]")
        self._h = C.c_void_p()
        _check(L.rt_overlay_mesh_create(v.ctypes.data, len(v), i.ctypes.data, len(i), t.ctypes.data, t.shape[1], t.shape[0], device, C.byref(self._h)))

    def draw(self, color, depth24, view_proj, worlds, world_its, light_dir, la, ld, stream: int | None = None, band=None, frame_height: int | None = None):
        """rt_overlay_draw: the draws of App.cpp:147-164 over `color` ([h, w, 4] float32) and `depth24` ([h, w], 24-bit values), in
        place.  Both numpy arrays (C-contiguous float32 / uint32) or both contiguous GPU tensors (depth24 int32 holding the values).
        Matrices are column-major sequences of 16 floats.  band = (band_rows, band_parts, band_part, compact): rt_overlay_draw_band on one
        owner's rows; with compact the planes hold only those rows and frame_height gives the frame's."""
        tensors = _is_tensor(color)
        if tensors:
            import torch
            h, w = depth24.shape
            if stream is None:
                stream = torch.cuda.current_stream().cuda_stream
        else:
            if not (isinstance(color, np.ndarray) and color.dtype == np.float32 and color.flags.c_contiguous and isinstance(depth24, np.ndarray) and depth24.dtype == np.uint32 and depth24.flags.c_contiguous):
                raise ValueError("color must be a C-contiguous float32 array and depth24 a C-contiguous uint32 array (they are updated in place)")
            h, w = depth24.shape
        pc, mem, _k1 = _plane_ptr(color, np.float32, 4 * h * w, "color")
        pz, m2, _k2 = _plane_ptr(depth24, np.uint32, h * w, "depth24")
        if mem != m2:
            raise ValueError("both planes must live in the same memory space")
        wm = np.ascontiguousarray(np.asarray(worlds, np.float32).reshape(-1, 16)); wi = np.ascontiguousarray(np.asarray(world_its, np.float32).reshape(-1, 16))
        if len(wm) != len(wi):
            raise ValueError("one worldIT per world matrix")
        p = rt_overlay_pass()
        p.view_proj[:] = [float(x) for x in np.asarray(view_proj, np.float32).reshape(16)]
        p.light_dir[:] = [float(x) for x in np.asarray(light_dir, np.float32)]; p.la[:] = [float(x) for x in np.asarray(la, np.float32)]; p.ld[:] = [float(x) for x in np.asarray(ld, np.float32)]
        p.n_draws = len(wm); p.world = wm.ctypes.data; p.world_it = wi.ctypes.data
        if band is None:
            _check(lib().rt_overlay_draw(self._h, C.byref(p), pc, pz, w, h, mem, stream))
        else:
            _check(lib().rt_overlay_draw_band(self._h, C.byref(p), pc, pz, w, frame_height if frame_height is not None else h, band[0], band[1], band[2], int(bool(band[3])), mem, stream))
        return color, depth24

    def close(self):
        if self._h:
            lib().rt_overlay_mesh_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


IDENTITY = np.eye(4, dtype=np.float32)


def scale_matrix(s: float) -> np.ndarray:
    m = np.eye(4, dtype=np.float32)
    m[0, 0] = m[1, 1] = m[2, 2] = np.float32(s)
    return m


class Mesh:
    """Mesh(filename, material_id, transform) -- Mesh.hpp:11-17.  `transform` is a [column][row] 4x4."""

    def __init__(self, filename=None, material_id: int = 0, transform=IDENTITY, *, positions=None, normals=None, faces=None):
        T = np.ascontiguousarray(transform, np.float32).reshape(16)
        m = rt_mesh()
        if filename is not None:
            _check(lib().rt_mesh_load_obj(os.fsencode(filename), material_id, T.ctypes.data, C.byref(m)))
        else:
            p = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
            f = np.ascontiguousarray(faces, np.uint32).reshape(-1, 3)
            n = None if normals is None else np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
            _check(lib().rt_mesh_from_arrays(p.ctypes.data, None if n is None else n.ctypes.data, len(p), f.ctypes.data, len(f), material_id, T.ctypes.data, C.byref(m)))
        try:
            self.m_vertecies = np.ctypeslib.as_array(m.vertices, (m.n_vert, 4)).copy() if m.n_vert else np.zeros((0, 4), np.float32)
            self.m_normals = np.ctypeslib.as_array(m.normals, (m.n_vert, 4)).copy() if m.n_vert else np.zeros((0, 4), np.float32)
            self.m_triangles = np.ctypeslib.as_array(m.triangles, (m.n_tri, 4)).copy() if m.n_tri else np.zeros((0, 4), np.uint32)
        finally:
            lib().rt_mesh_free(C.byref(m))

    @staticmethod
    def from_buffers(vertices, normals, triangles) -> "Mesh":
        """Wrap already-built [n,4] / [n,4] / [m,4] arrays."""
        me = Mesh.__new__(Mesh)
        me.m_vertecies = np.ascontiguousarray(vertices, np.float32).reshape(-1, 4)
        me.m_normals = np.ascontiguousarray(normals, np.float32).reshape(-1, 4)
        me.m_triangles = np.ascontiguousarray(triangles, np.uint32).reshape(-1, 4)
        return me


def _mesh_array(meshes):
    """The meshes as the C ABI's rt_mesh array, and the numpy arrays its pointers refer to (to be kept alive by the caller)."""
    arr = (rt_mesh * max(1, len(meshes)))()
    keep = []
    for i, me in enumerate(meshes):
        v, n, t = (np.ascontiguousarray(me.m_vertecies, np.float32), np.ascontiguousarray(me.m_normals, np.float32), np.ascontiguousarray(me.m_triangles, np.uint32))
        keep += [v, n, t]
        arr[i].vertices = v.ctypes.data_as(C.POINTER(C.c_float)); arr[i].normals = n.ctypes.data_as(C.POINTER(C.c_float))
        arr[i].triangles = t.ctypes.data_as(C.POINTER(C.c_uint32)); arr[i].n_vert = len(v); arr[i].n_tri = len(t)
    return arr, keep


class Octree:
    """Octree(meshes, depth_limit, max_triangles_per_node, max_triangles_per_leaf, keep_after) -- Octree.hpp:27-36."""

    def __init__(self, meshes, depth_limit: int = 18, max_triangles_per_node: int = 10, max_triangles_per_leaf: int = 6,
                 keep_triangles_after_this_many_overlaps: int = 6, *, device: int | None = None):
        """device=None: the host builder; device=i: the same construction with the overlap tests on GPU i (identical arrays)."""
        arr, keep = _mesh_array(meshes)
        o = rt_octree()
        if device is None:
            _check(lib().rt_octree_build(arr, len(meshes), depth_limit, max_triangles_per_node, max_triangles_per_leaf, keep_triangles_after_this_many_overlaps, C.byref(o)))
        else:
            _check(lib().rt_octree_build_device(arr, len(meshes), depth_limit, max_triangles_per_node, max_triangles_per_leaf, keep_triangles_after_this_many_overlaps, int(device), C.byref(o)))
        try:
            self.m_vertecies = np.ctypeslib.as_array(o.vertices, (o.n_vert, 4)).copy() if o.n_vert else np.zeros((0, 4), np.float32)
            self.m_normals = np.ctypeslib.as_array(o.normals, (o.n_vert, 4)).copy() if o.n_vert else np.zeros((0, 4), np.float32)
            self.m_compressed_triangles = np.ctypeslib.as_array(o.tri_refs, (o.n_ref, 4)).copy() if o.n_ref else np.zeros((0, 4), np.uint32)
            self.m_compressed_node_buffer = np.ctypeslib.as_array(o.nodes, (o.n_words,)).copy() if o.n_words else np.zeros((0,), np.uint32)
            self._min = np.array(o.bounds_min[:], np.float32)
            self._max = np.array(o.bounds_max[:], np.float32)
            self.max_depth = int(o.max_depth)
        finally:
            lib().rt_octree_free(C.byref(o))

    def GetMinBounds(self) -> np.ndarray:
        return self._min

    def GetMaxBounds(self) -> np.ndarray:
        return self._max

    def as_dict(self) -> dict:
        return dict(vertices=self.m_vertecies, normals=self.m_normals, tri_refs=self.m_compressed_triangles, nodes=self.m_compressed_node_buffer,
                    bounds_min=self._min, bounds_max=self._max)


class Scene:
    """Device-resident scene: the four buffers of App.cpp:23-26 plus the octree bounds and the skybox."""

    def __init__(self, vertices, normals, tri_refs, nodes, bounds_min, bounds_max, device: int | None = None):
        v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 4)
        n = np.ascontiguousarray(normals, np.float32).reshape(-1, 4)
        t = np.ascontiguousarray(tri_refs, np.uint32).reshape(-1, 4)
        k = np.ascontiguousarray(nodes, np.uint32).ravel()
        if len(n) != len(v):
            raise RtError(RT_EINVAL, "vertices and normals differ in length")
        lo = np.ascontiguousarray(bounds_min, np.float32); hi = np.ascontiguousarray(bounds_max, np.float32)
        if device is not None:
            _check(lib().rt_set_device(device))
        h = C.c_void_p()
        _check(lib().rt_scene_create(v.ctypes.data, n.ctypes.data, len(v), t.ctypes.data, len(t), k.ctypes.data, len(k), lo.ctypes.data, hi.ctypes.data, C.byref(h)))
        self._h = h
        self.n_ref, self.n_words = len(t), len(k)

    @staticmethod
    def from_meshes(meshes, depth_limit: int = 18, max_triangles_per_node: int = 10, max_triangles_per_leaf: int = 6,
                    keep_triangles_after_this_many_overlaps: int = 6, *, device: int = 0) -> "Scene":
        """Octree(meshes, ...) and the uploads of App.cpp:22-26 in one call that stays on the device (rt_scene_create_from_meshes);
        the octree's bounds, which the uniforms need, are left in `bounds_min` / `bounds_max`."""
        arr, keep = _mesh_array(meshes)
        lo, hi = np.zeros(3, np.float32), np.zeros(3, np.float32)
        h = C.c_void_p()
        _check(lib().rt_scene_create_from_meshes(arr, len(meshes), depth_limit, max_triangles_per_node, max_triangles_per_leaf, keep_triangles_after_this_many_overlaps,
                                                  int(device), lo.ctypes.data, hi.ctypes.data, C.byref(h)))
        del keep
        sc = Scene.__new__(Scene)
        sc._h = h
        sc.n_ref = sc.n_words = None
        sc.bounds_min, sc.bounds_max = lo, hi
        return sc

    @staticmethod
    def from_octree(octree: Octree, device: int | None = None) -> "Scene":
        return Scene(octree.m_vertecies, octree.m_normals, octree.m_compressed_triangles, octree.m_compressed_node_buffer,
                     octree.GetMinBounds(), octree.GetMaxBounds(), device)

    @staticmethod
    def from_dict(d: dict, device: int | None = None) -> "Scene":
        return Scene(d["vertices"], d["normals"], d["tri_refs"], d["nodes"], d["bounds_min"], d["bounds_max"], device)

    def set_skybox(self, faces: np.ndarray):
        """faces: uint8 [6, h, w, 4] in the order +x,-x,+y,-y,+z,-z (Skybox.cpp:11-18)."""
        f = np.ascontiguousarray(faces, np.uint8)
        if f.ndim != 4 or f.shape[0] != 6 or f.shape[3] != 4:
            raise RtError(RT_EINVAL, "skybox must be uint8 [6, h, w, 4]")
        ptrs = (C.c_void_p * 6)(*[f[i].ctypes.data for i in range(6)])
        _check(lib().rt_skybox_set(self._h, ptrs, f.shape[2], f.shape[1]))

    def set_counting(self, enabled: bool):
        _check(lib().rt_scene_set_counting(self._h, int(enabled)))

    def set_kernel_timing(self, enabled: bool):
        _check(lib().rt_scene_set_kernel_timing(self._h, int(enabled)))

    def kernel_time(self):
        """(summed ms, launches) of the dominant kernel since the last read; synchronises the device."""
        ms, n = C.c_double(), C.c_uint32()
        _check(lib().rt_scene_kernel_time(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def set_pipeline(self, name: str):
        """'auto' (default: the fastest of the others, measured), 'megakernel', 'wavefront' or 'persistent'; results are identical."""
        _check(lib().rt_scene_set_pipeline(self._h, {"megakernel": 0, "wavefront": 1, "persistent": 2, "auto": 3}[name]))

    def pipeline_state(self):
        """(name of the pipeline the next frame runs with, {name: measured ms per frame or None})."""
        sel, ms = C.c_int(), (C.c_double * 3)()
        _check(lib().rt_scene_pipeline_state(self._h, C.byref(sel), ms))
        names = ["megakernel", "wavefront", "persistent"]
        return names[sel.value], {n: (ms[i] if ms[i] >= 0 else None) for i, n in enumerate(names)}

    def set_sequence_batch(self, max_frames: int):
        _check(lib().rt_scene_set_sequence_batch(self._h, max_frames))

    def set_traversal_order(self, order: str):
        """'auto' (default: bit-identical outputs by whichever of the next two the scene measures to be faster), 'reference' (the
        shader's own child order), 'near_first_exact' (near first, with the shader's result: see rt_hotpath.h) or 'near_first'
        (faster, held to the colour tolerance only)."""
        _check(lib().rt_scene_set_traversal_order(self._h, ORDER_NAMES.index(order)))

    def selection_state(self):
        """(pipeline, order, {candidate: measured ms per frame or None}) of what automatic selection would run next."""
        p, o, ms = C.c_int(), C.c_int(), (C.c_double * 5)()
        _check(lib().rt_scene_selection_state(self._h, C.byref(p), C.byref(o), ms))
        return PIPELINE_NAMES[p.value], ORDER_NAMES[o.value], {n: (ms[i] if ms[i] >= 0 else None) for i, n in enumerate(CANDIDATE_NAMES)}

    def last_mode(self):
        """(pipeline, order) the last render call actually ran with."""
        p, o = C.c_int(), C.c_int()
        _check(lib().rt_scene_last_mode(self._h, C.byref(p), C.byref(o)))
        return PIPELINE_NAMES[p.value], ORDER_NAMES[o.value]

    def set_subframes(self, n: int):
        _check(lib().rt_scene_set_subframes(self._h, n))

    def set_frames_in_flight(self, n: int):
        _check(lib().rt_scene_set_frames_in_flight(self._h, n))

    def set_frames_per_launch(self, n: int):
        _check(lib().rt_scene_set_frames_per_launch(self._h, n))

    def stats(self) -> dict:
        s = rt_stats()
        _check(lib().rt_scene_stats(self._h, C.byref(s)))
        return {n: int(getattr(s, n)) for n, _ in rt_stats._fields_}

    def render_raw(self, frame: rt_frame, target: rt_target):
        _check(lib().rt_render(self._h, C.byref(frame), C.byref(target)))

    def render_noalloc_raw(self, frame: rt_frame, target: rt_target):
        """rt_render_noalloc: rt_render on a device-memory target without any allocation (static device scratch)."""
        _check(lib().rt_render_noalloc(self._h, C.byref(frame), C.byref(target)))

    def render_sequence_raw(self, frames, target: rt_target):
        """Consecutive accumulation frames into one target in one call (frames: list of rt_frame)."""
        arr = (rt_frame * len(frames))(*frames)
        _check(lib().rt_render_sequence(self._h, arr, len(frames), C.byref(target)))

    def render(self, uniforms, width: int, height: int, *, accum_index: int = 1, blend: bool = True, color: np.ndarray | None = None,
               planes: bool = True, ndc=None, band=(16, 1, 0)) -> dict:
        """Render one frame into host arrays (convenience path used by tests).  Returns dict(rgba, depth[, kind, slot, t, nrays])."""
        u = uniforms.pack() if isinstance(uniforms, Uniforms) else np.ascontiguousarray(uniforms, np.float32)
        fr = make_frame(u, width, height, accum_index=accum_index, blend=blend, band=band)
        out = {"rgba": np.zeros((height, width, 4), np.float32) if color is None else np.ascontiguousarray(color, np.float32), "depth": np.zeros((height, width), np.float32)}
        tg = rt_target()
        tg.memory, tg.sync = RT_MEM_HOST, 1
        tg.color, tg.depth = out["rgba"].ctypes.data, out["depth"].ctypes.data
        if planes:
            out["kind"] = np.zeros((height, width), np.uint32); tg.hit_kind = out["kind"].ctypes.data
            out["slot"] = np.zeros((height, width), np.uint32); tg.hit_slot = out["slot"].ctypes.data
            out["t"] = np.zeros((height, width), np.float32); tg.hit_t = out["t"].ctypes.data
            out["nrays"] = np.zeros((height, width), np.uint32); tg.n_rays = out["nrays"].ctypes.data
        if ndc is not None:
            nx, ny = (np.ascontiguousarray(a, np.float32) for a in ndc)
            assert nx.shape == (height, width) and ny.shape == (height, width)
            fr.ndc_x, fr.ndc_y = nx.ctypes.data, ny.ctypes.data
        self.render_raw(fr, tg)
        return out

    def close(self):
        if getattr(self, "_h", None):
            lib().rt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def band_row_count(frame: rt_frame) -> int:
    return int(lib().rt_band_row_count(C.byref(frame)))


def make_frame(uniforms70: np.ndarray, width: int, height: int, *, accum_index: int = 1, blend: bool = True, band=(16, 1, 0), stream: int | None = None, compact: bool = False) -> rt_frame:
    u = np.ascontiguousarray(uniforms70, np.float32)
    if u.size != 70:
        raise RtError(RT_EINVAL, "uniform block must have 70 words")
    fr = rt_frame()
    C.memmove(C.byref(fr.u), u.ctypes.data, 280)
    fr.width, fr.height, fr.accum_index, fr.blend = width, height, accum_index, int(blend)
    fr.band_rows, fr.band_parts, fr.band_part = band
    fr.compact_rows = int(compact)
    fr.stream = stream
    return fr


class RenderPass:
    """The ray-tracing draw of App::Render with its progressive accumulation state, on device memory.

    `color` is a device pointer to a [height][width][4] float32 target that persists across frames
    (Framebuffer.cpp:6); `camera_changed()` resets the still-frame counter like App.cpp:90-92,209.
    """

    def __init__(self, scene: Scene, width: int, height: int, color_ptr: int, *, depth_ptr: int | None = None, band=(16, 1, 0), stream: int | None = None):
        self.scene, self.width, self.height = scene, width, height
        self.color_ptr, self.depth_ptr, self.band, self.stream = color_ptr, depth_ptr, band, stream
        self.m_still_frame_counter = 1

    def camera_changed(self):
        self.m_still_frame_counter = 1

    def Render(self, uniforms, sync: bool = False):
        u = uniforms.pack() if isinstance(uniforms, Uniforms) else uniforms
        fr = make_frame(u, self.width, self.height, accum_index=self.m_still_frame_counter, blend=True, band=self.band, stream=self.stream)
        self.m_still_frame_counter += 1            # App.cpp:105
        tg = rt_target()
        tg.memory, tg.sync = RT_MEM_DEVICE, int(sync)
        tg.color, tg.depth = self.color_ptr, self.depth_ptr
        self.scene.render_raw(fr, tg)

    def RenderFrames(self, uniforms_per_frame, sync: bool = False):
        """len(uniforms_per_frame) consecutive Render calls (camera unchanged, time advancing) issued as one sequence."""
        frames = []
        for uniforms in uniforms_per_frame:
            u = uniforms.pack() if isinstance(uniforms, Uniforms) else uniforms
            frames.append(make_frame(u, self.width, self.height, accum_index=self.m_still_frame_counter, blend=True, band=self.band, stream=self.stream))
            self.m_still_frame_counter += 1
        tg = rt_target()
        tg.memory, tg.sync = RT_MEM_DEVICE, int(sync)
        tg.color, tg.depth = self.color_ptr, self.depth_ptr
        self.scene.render_sequence_raw(frames, tg)
