"""The C++ host mirror (raytracer_amd/host/rt_host.hpp) driven by its demo program: the reference-named Mesh / Octree
classes give the arrays the Python mirror gives (same C ABI underneath), and RayTracePass accumulates frames in device
memory to the same bits as Scene.render."""
import os
import subprocess

import numpy as np
import pytest

from raytracer_amd import api
from raytracer_amd.camera import Camera
from raytracer_amd.params import Uniforms

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "raytracer_amd", "host", "host_demo")


def _soup(seed=5, n_tri=400):
    rng = np.random.default_rng(seed)
    centres = rng.uniform(-0.6, 0.6, (n_tri, 1, 3)) + np.array([0.0, 0.75, 0.0])
    pos = (centres + rng.uniform(-0.08, 0.08, (n_tri, 3, 3))).reshape(-1, 3).astype(np.float32)
    faces = np.arange(3 * n_tri, dtype=np.uint32).reshape(-1, 3)
    return pos, faces


def _run_demo(tmp_path, pos, faces, render=None, overlay=None):
    mesh_bin, out_bin = tmp_path / "mesh.bin", tmp_path / "out.bin"
    with open(mesh_bin, "wb") as f:
        f.write(np.array([len(pos), len(faces)], np.uint32).tobytes()); f.write(pos.tobytes()); f.write(faces.tobytes())
    cmd = [DEMO, str(mesh_bin), str(out_bin)]
    walk = bool(render) and render[0] == "walk"
    if walk:
        cmd += ["walk", str(render[1]), str(render[2]), str(render[3])]
    elif render:
        us, w, h = render
        (tmp_path / "u.bin").write_bytes(np.concatenate(us).astype(np.float32).tobytes())
        cmd += [str(tmp_path / "u.bin"), str(w), str(h), str(len(us))]
    if overlay:
        ov, oi, ot = overlay
        with open(tmp_path / "overlay.bin", "wb") as f:
            f.write(np.array([len(ov), len(oi), ot.shape[1], ot.shape[0]], np.uint32).tobytes()); f.write(ov.tobytes()); f.write(oi.tobytes()); f.write(ot.tobytes())
        cmd += [str(tmp_path / "overlay.bin")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    raw = out_bin.read_bytes()
    nv, nr, nw = np.frombuffer(raw, np.uint64, 3)
    off = 24
    def take(dtype, count):
        nonlocal off
        a = np.frombuffer(raw, dtype, int(count), off); off += a.nbytes
        return a
    d = dict(bounds_min=take(np.float32, 3), bounds_max=take(np.float32, 3), vertices=take(np.float32, nv * 4).reshape(-1, 4), normals=take(np.float32, nv * 4).reshape(-1, 4),
             tri_refs=take(np.uint32, nr * 4).reshape(-1, 4), nodes=take(np.uint32, nw))
    if render:
        w, h = render[1], render[2]
        d["rgba"] = take(np.float32, w * h * 4).reshape(h, w, 4); d["depth"] = take(np.float32, w * h).reshape(h, w)
    if walk:
        d["uniforms"] = take(np.float32, render[3] * 70).reshape(-1, 70)
    if overlay:
        d["depth24"] = take(np.uint32, render[1] * render[2]).reshape(render[2], render[1])
    assert off == len(raw)
    return d


def _python_octree(pos, faces):
    mesh = api.Mesh(positions=pos, faces=faces, material_id=1, transform=api.scale_matrix(2.0))
    return api.Octree([mesh], 18, 10, 6, 6)


def test_cpp_octree_equals_python_mirror(tmp_path):
    pos, faces = _soup()
    d = _run_demo(tmp_path, pos, faces)
    o = _python_octree(pos, faces).as_dict()
    for k in ("vertices", "normals", "bounds_min", "bounds_max"):
        assert np.array_equal(d[k].view(np.uint32), np.ascontiguousarray(o[k], np.float32).view(np.uint32).reshape(d[k].shape)), k
    assert np.array_equal(d["tri_refs"], o["tri_refs"].reshape(-1, 4)) and np.array_equal(d["nodes"], o["nodes"])
    assert len(d["nodes"]) > 100


@pytest.mark.gpu
def test_cpp_render_pass_accumulates_like_scene_render(tmp_path):
    pos, faces = _soup()
    w, h, n = 160, 96, 3
    us = [Uniforms.from_camera(Camera(aspect=w / h), time=k / 60.0).pack() for k in range(n)]
    d = _run_demo(tmp_path, pos, faces, render=(us, w, h))
    sc = api.Scene.from_octree(_python_octree(pos, faces))   # no skybox in either: texture() returns 0
    acc, depth = None, None
    for k in range(n):
        r = sc.render(us[k], w, h, accum_index=k + 1, color=acc, planes=False)
        acc, depth = r["rgba"], r["depth"]
    sc.close()
    assert np.array_equal(d["rgba"].view(np.uint32), acc.view(np.uint32))
    assert np.array_equal(d["depth"].view(np.uint32), depth.view(np.uint32))


@pytest.mark.gpu
def test_cpp_camera_walk_like_the_reference_loop(tmp_path):
    """The demo's `walk` mode plays App's loop with the C++ Camera / CameraManipulator mirror (W held, then released for
    the last two frames): its uniforms equal the Python mirror's walk, and the image it accumulates equals the frames of
    the standing camera rendered through Scene.render and through the CPU oracle."""
    from oracle import oracle
    from raytracer_amd import glm_camera as g
    pos, faces = _soup()
    w, h, n = 160, 96, 6
    d = _run_demo(tmp_path, pos, faces, render=("walk", w, h, n))
    cam = g.Camera(); cam.SetView((2, 2, 2), (0, 0, 0), (0, 1, 0)); cam.SetAspect(np.float32(w) / np.float32(h))
    man = g.CameraManipulator(); man.SetCamera(cam)
    p1, p2, pw, ph = g.default_portals()
    man.KeyboardDown(g.KEY_W)
    us, restarts = [], []
    for k in range(n):
        if k + 2 == n:
            man.KeyboardUp(g.KEY_W)
        restarts.append(man.Update(np.float32(1.0) / np.float32(60.0), p1, p2, pw, ph))
        us.append(g.frame_uniforms(cam, p1, p2, pw, ph, time=np.float32(k) / np.float32(60.0)))
    assert np.array_equal(np.stack(us).view(np.uint32), d["uniforms"].view(np.uint32))
    assert restarts[:n - 2] == [True] * (n - 2)
    eyes = np.stack(us)[:, 16:19]
    assert np.allclose(np.linalg.norm(eyes[1] - eyes[0]), 16.0 / 60.0, atol=1e-4) and not np.array_equal(eyes[n - 3], eyes[0])
    first = max(k for k in range(n) if restarts[k])          # accumulation restarts there
    o = _python_octree(pos, faces)
    sc = api.Scene.from_octree(o)
    osc = oracle.OracleScene(o.as_dict(), None)
    acc, depth, oacc = None, None, np.zeros((h, w, 4), np.float32)
    for i, k in enumerate(range(first, n)):
        r = sc.render(us[k], w, h, accum_index=i + 1, color=acc, planes=False)
        acc, depth = r["rgba"], r["depth"]
        oacc = osc.render(us[k], w, h, accum_index=i + 1, rgba=oacc, taps=False)["rgba"]
    sc.close()
    assert np.array_equal(d["rgba"].view(np.uint32), acc.view(np.uint32))
    assert np.array_equal(d["depth"].view(np.uint32), depth.view(np.uint32))
    assert np.array_equal(d["rgba"].view(np.uint32), oacc.view(np.uint32))


@pytest.mark.gpu
def test_cpp_whole_frame_with_the_raster_pass(tmp_path):
    """The demo's walk with rt::RasterPass after every ray-tracing draw (the whole of App::Render) against the Python mirror of the
    same loop: accumulating rt_render, rt_depth_to_attachment, rt_overlay_draw with viewProj from the camera mirror."""
    from raytracer_amd import glm_camera as g
    pos, faces = _soup()
    rng = np.random.default_rng(8)
    n_t = 60                       # a small mesh in the default view: instance (0, -1) is not translated
    ov = np.zeros((3 * n_t, 8), np.float32)
    ov[:, :3] = (rng.uniform(0.2, 1.2, (n_t, 1, 3)) + rng.uniform(-0.3, 0.3, (n_t, 3, 3))).reshape(-1, 3)      # between the eye and the point it looks at
    ov[:, 3:6] = rng.uniform(-1, 1, (3 * n_t, 3)); ov[:, 6:8] = rng.uniform(0, 1, (3 * n_t, 2))
    oi = np.arange(3 * n_t, dtype=np.uint32)
    ot = rng.integers(0, 256, (8, 16, 4), dtype=np.uint8)
    w, h, n = 160, 96, 5
    d = _run_demo(tmp_path, pos, faces, render=("walk", w, h, n), overlay=(ov, oi, ot))
    cam = g.Camera(); cam.SetView((2, 2, 2), (0, 0, 0), (0, 1, 0)); cam.SetAspect(np.float32(w) / np.float32(h))
    man = g.CameraManipulator(); man.SetCamera(cam)
    p1, p2, pw, ph = g.default_portals()
    man.KeyboardDown(g.KEY_W)
    sc = api.Scene.from_octree(_python_octree(pos, faces))
    mesh = api.OverlayMesh(ov, oi, ot)
    worlds, its = [], []
    for i in (-1, 0, 1):
        for j in (-1, 0, 1):
            t = np.array([4.0 * i, 4.0 + 4.0 * j, 0.0], np.float32)
            wm = np.eye(4, dtype=np.float32); wm[3, :3] = t
            it = np.eye(4, dtype=np.float32); it[:3, 3] = -t
            worlds.append(wm.reshape(16)); its.append(it.reshape(16))
    acc, counter = None, 1
    for k in range(n):
        if k + 2 == n:
            man.KeyboardUp(g.KEY_W)
        if man.Update(np.float32(1.0) / np.float32(60.0), p1, p2, pw, ph):
            counter = 1
        u = g.frame_uniforms(cam, p1, p2, pw, ph, time=np.float32(k) / np.float32(60.0))
        r = sc.render(u, w, h, accum_index=counter, color=acc, planes=False)
        counter += 1
        acc = np.ascontiguousarray(r["rgba"], np.float32); d24 = api.depth_to_attachment(r["depth"])
        mesh.draw(acc, d24, np.asarray(cam.GetViewProj(), np.float32).reshape(16), worlds, its, (-1.0, -2.0, -0.5), (0, 0, 0), (1, 1, 1))
    sc.close(); mesh.close()
    assert (d24 != api.depth_to_attachment(r["depth"])).sum() > 200          # the mesh is in view
    assert np.array_equal(d["depth24"], d24)
    assert np.array_equal(d["rgba"].view(np.uint32), acc.view(np.uint32))
