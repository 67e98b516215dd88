"""The multi-GPU functions of the C ABI (include/rt_dist.h) on the one GPU of the test box: a communicator of one rank (RCCL does
not accept two ranks on one device), the scene and skybox created through the broadcast path, a band-compact render and the gather
into the frame -- against a plain rt_scene_create / rt_render of the same frame.  The band arithmetic for several ranks is the same
as raytracer_amd/dist.py's, which tests/test_dist_cpu.py checks with two and three ranks."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import api, scenes
from raytracer_amd.camera import Camera
from raytracer_amd.params import Uniforms

pytestmark = pytest.mark.gpu


def test_single_rank_communicator_scene_broadcast_and_gather(test_sky):
    import torch
    L = api.lib()
    L.rt_dist_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.rt_dist_scene_create.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    L.rt_dist_skybox_set.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int]
    L.rt_dist_frame_bands.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(api.rt_frame)]
    L.rt_dist_gather_frame.argtypes = [C.c_void_p, C.c_int, C.POINTER(api.rt_frame), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.rt_dist_destroy.argtypes = [C.c_void_p]
    L.rt_dist_rank.argtypes = [C.c_void_p]
    L.rt_dist_world_size.argtypes = [C.c_void_p]
    uid = (C.c_uint8 * 128)()
    assert L.rt_dist_make_id(uid) == api.RT_OK, L.rt_last_error()
    d = C.c_void_p()
    assert L.rt_dist_init(uid, 1, 0, 0, C.byref(d)) == api.RT_OK, L.rt_last_error()
    assert L.rt_dist_rank(d) == 0 and L.rt_dist_world_size(d) == 1
    oc = scenes.small_test_scene()
    v, n, t, k = (np.ascontiguousarray(a) for a in (oc.m_vertecies, oc.m_normals, oc.m_compressed_triangles, oc.m_compressed_node_buffer))
    lo, hi = np.ascontiguousarray(oc.GetMinBounds()), np.ascontiguousarray(oc.GetMaxBounds())
    h = C.c_void_p()
    assert L.rt_dist_scene_create(d, 0, v.ctypes.data, n.ctypes.data, len(v), t.ctypes.data, len(t), k.ctypes.data, len(k), lo.ctypes.data, hi.ctypes.data, C.byref(h)) == api.RT_OK, L.rt_last_error()
    sky = np.ascontiguousarray(test_sky)
    faces = (C.c_void_p * 6)(*[sky[i].ctypes.data for i in range(6)])
    assert L.rt_dist_skybox_set(d, 0, h, faces, sky.shape[2], sky.shape[1]) == api.RT_OK, L.rt_last_error()
    w, ht = 200, 150                          # the last 16-row strip is clipped by the frame
    u = Uniforms.from_camera(Camera(eye=(2.5, 2.5, 2.5), at=(0.0, 1.5, 0.0), aspect=w / ht)).pack()
    fr = api.make_frame(u, w, ht, stream=torch.cuda.current_stream().cuda_stream)
    assert L.rt_dist_frame_bands(d, 16, C.byref(fr)) == api.RT_OK
    assert (fr.band_rows, fr.band_parts, fr.band_part, fr.compact_rows) == (16, 1, 0, 1)
    rows = api.band_row_count(fr)
    band = torch.zeros((rows, w, 4), device="cuda")
    depth_band = torch.zeros((rows, w), device="cuda")
    tg = api.rt_target()
    tg.memory, tg.sync, tg.color, tg.depth = api.RT_MEM_DEVICE, 0, band.data_ptr(), depth_band.data_ptr()
    assert L.rt_render(h, C.byref(fr), C.byref(tg)) == api.RT_OK, L.rt_last_error()
    frame = torch.full((ht, w, 4), float("nan"), device="cuda")
    depth = torch.full((ht, w), float("nan"), device="cuda")
    assert L.rt_dist_gather_frame(d, 0, C.byref(fr), band.data_ptr(), frame.data_ptr(), 16, fr.stream) == api.RT_OK, L.rt_last_error()
    assert L.rt_dist_gather_frame(d, 0, C.byref(fr), depth_band.data_ptr(), depth.data_ptr(), 4, fr.stream) == api.RT_OK, L.rt_last_error()
    torch.cuda.synchronize()
    ref_scene = api.Scene.from_octree(oc, device=0)
    ref_scene.set_skybox(test_sky)
    ref = ref_scene.render(u, w, ht, planes=False)
    assert np.array_equal(frame.cpu().numpy().view(np.uint32), ref["rgba"].view(np.uint32))
    assert np.array_equal(depth.cpu().numpy().view(np.uint32), ref["depth"].view(np.uint32))
    # a frame whose band selection is not the communicator's is refused
    bad = api.make_frame(u, w, ht, band=(16, 2, 0), compact=True)
    assert L.rt_dist_gather_frame(d, 0, C.byref(bad), band.data_ptr(), frame.data_ptr(), 16, None) == api.RT_EINVAL
    ref_scene.close()
    assert L.rt_scene_destroy(h) == api.RT_OK
    assert L.rt_dist_destroy(d) == api.RT_OK


def test_two_ranks_on_two_devices(tmp_path, test_sky):
    """Two processes, one per GPU, through rt_dist_scene_create / rt_dist_skybox_set / rt_dist_gather_frame over RCCL: the frame
    the root gathers equals the single-GPU frame bit for bit.  Needs two devices (RCCL does not accept two ranks on one)."""
    import os
    import subprocess
    import sys
    if api.device_count() < 2:
        pytest.skip("needs two GPUs")
    here = os.path.dirname(os.path.abspath(__file__))
    out, idf = str(tmp_path / "frame.npy"), str(tmp_path / "id")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, os.path.join(here, "dist_cabi_child.py"), str(r), "2", idf, out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    logs = [p.communicate(timeout=300)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), logs
    got = np.load(out)
    w, h = 200, 150
    sc = api.Scene.from_octree(scenes.small_test_scene(), device=0)
    sc.set_skybox(test_sky)
    acc = None
    for k in range(2):
        u = Uniforms.from_camera(Camera(eye=(2.5, 2.5, 2.5), at=(0.0, 1.5, 0.0), aspect=w / h), time=k / 60.0).pack()
        r = sc.render(u, w, h, accum_index=k + 1, color=acc, planes=False)
        acc = r["rgba"]
    assert np.array_equal(got[:, :w * 4].reshape(h, w, 4).view(np.uint32), acc.view(np.uint32))
    assert np.array_equal(np.ascontiguousarray(got[:, w * 4:]).view(np.uint32), r["depth"].view(np.uint32))
    sc.close()
