"""ctypes mirror of include/rt_dist.h: the multi-GPU side of the C ABI (scene broadcast and band gather over RCCL behind
`librt_hotpath.so`), for hosts that drive several ranks without torch.distributed in the data path.

No counterpart in the reference (one process, one GL context: /root/reference/app/main.cpp:19-174).  The id that the ranks
join with travels by whatever means the caller has; `exchange_id_by_file` is the simplest one for ranks of one machine.
"""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np

from . import api

ID_BYTES = 128


def _lib():
    L = api.lib()
    L.rt_dist_make_id.argtypes = [C.c_void_p]
    L.rt_dist_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.rt_dist_scene_create.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    L.rt_dist_skybox_set.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int]
    L.rt_dist_frame_bands.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(api.rt_frame)]
    L.rt_dist_gather_frame.argtypes = [C.c_void_p, C.c_int, C.POINTER(api.rt_frame), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.rt_dist_destroy.argtypes = [C.c_void_p]
    L.rt_dist_rank.argtypes = [C.c_void_p]
    L.rt_dist_world_size.argtypes = [C.c_void_p]
    return L


def make_id() -> bytes:
    uid = (C.c_uint8 * ID_BYTES)()
    api._check(_lib().rt_dist_make_id(uid))
    return bytes(uid)


def exchange_id_by_file(path: str, rank: int, root: int = 0, timeout_s: float = 120.0) -> bytes:
    """The root writes a fresh id to `path` (atomically: written beside it and renamed); the others wait for the file."""
    if rank == root:
        uid = make_id()
        tmp = path + ".tmp%d" % os.getpid()
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)
        return uid
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout_s:
            raise TimeoutError("no communicator id at " + path)
        time.sleep(0.02)
    with open(path, "rb") as f:
        uid = f.read()
    if len(uid) != ID_BYTES:
        raise ValueError("communicator id file has %d bytes" % len(uid))
    return uid


class DistComm:
    """rt_dist: one rank of a communicator of `world_size` processes, one per HIP device."""

    def __init__(self, uid: bytes, world_size: int, rank: int, device: int):
        self._L = _lib()
        self._h = C.c_void_p()
        buf = (C.c_uint8 * ID_BYTES).from_buffer_copy(uid)
        api._check(self._L.rt_dist_init(buf, world_size, rank, device, C.byref(self._h)))
        self.world_size, self.rank, self.device = world_size, rank, device

    def scene_create(self, arrays: dict | None, root: int = 0) -> "api.Scene":
        """rt_dist_scene_create: `arrays` (vertices, normals, tri_refs, nodes, bounds_min, bounds_max) on the root, None elsewhere."""
        h = C.c_void_p()
        if self.rank == root:
            v = np.ascontiguousarray(arrays["vertices"], np.float32).reshape(-1, 4); n = np.ascontiguousarray(arrays["normals"], np.float32).reshape(-1, 4)
            t = np.ascontiguousarray(arrays["tri_refs"], np.uint32).reshape(-1, 4); k = np.ascontiguousarray(arrays["nodes"], np.uint32).ravel()
            lo = np.ascontiguousarray(arrays["bounds_min"], np.float32); hi = np.ascontiguousarray(arrays["bounds_max"], np.float32)
            rc = self._L.rt_dist_scene_create(self._h, root, v.ctypes.data, n.ctypes.data, len(v), t.ctypes.data, len(t), k.ctypes.data, len(k), lo.ctypes.data, hi.ctypes.data, C.byref(h))
        else:
            rc = self._L.rt_dist_scene_create(self._h, root, None, None, 0, None, 0, None, 0, None, None, C.byref(h))
        api._check(rc)
        sc = api.Scene.__new__(api.Scene)
        sc._h = h
        sc.n_ref = sc.n_words = None
        return sc

    def skybox_set(self, scene: "api.Scene", faces: np.ndarray | None, root: int = 0):
        """rt_dist_skybox_set: `faces` uint8 [6, h, w, 4] on the root, None elsewhere."""
        if self.rank == root:
            f = np.ascontiguousarray(faces, np.uint8)
            ptrs = (C.c_void_p * 6)(*[f[i].ctypes.data for i in range(6)])
            api._check(self._L.rt_dist_skybox_set(self._h, root, scene._h, ptrs, f.shape[2], f.shape[1]))
        else:
            api._check(self._L.rt_dist_skybox_set(self._h, root, scene._h, None, 0, 0))

    def frame_bands(self, frame: "api.rt_frame", band_rows: int = 16) -> "api.rt_frame":
        api._check(self._L.rt_dist_frame_bands(self._h, band_rows, C.byref(frame)))
        return frame

    def gather_frame(self, frame: "api.rt_frame", band_ptr: int, frame_ptr: int | None, bytes_per_pixel: int, stream: int | None, root: int = 0):
        """rt_dist_gather_frame on device pointers; returns after the enqueue on `stream`."""
        api._check(self._L.rt_dist_gather_frame(self._h, root, C.byref(frame), band_ptr, frame_ptr, bytes_per_pixel, stream))

    def close(self):
        if getattr(self, "_h", None):
            self._L.rt_dist_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
