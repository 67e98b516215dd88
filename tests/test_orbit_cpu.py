"""CPU tests of the headless camera-path driver (raytracer_amd/orbit.py): the path itself, and the driver's pose /
accumulation / gather logic on two gloo ranks with the CPU oracle standing in for the device render pass."""
import math
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd import dist as rdist, orbit, scenes

CFG = dict(width=40, height=50, spp=3, n_poses=120, max_recursion_limit=3, band_rows=8)
POSES = [0, 37, 119]


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_orbit_path():
    cfg = orbit.OrbitConfig()
    assert (cfg.width, cfg.height, cfg.spp, cfg.n_poses, cfg.max_recursion_limit) == (7680, 4320, 64, 120, 5)
    assert np.allclose(orbit.orbit_camera(cfg, 0).eye, (3.0, 2.1, 0.0))
    assert np.allclose(orbit.orbit_camera(cfg, 30).eye, (0.0, 2.1, 3.0), atol=1e-12)
    assert np.allclose(orbit.orbit_camera(cfg, 120).eye, orbit.orbit_camera(cfg, 0).eye)
    blocks = orbit.pose_uniforms(orbit.OrbitConfig(spp=4), 7)
    assert [float(b[20]) for b in blocks] == [np.float32(k / 60.0) for k in range(4)]       # time_k = k / 60
    assert all(int(b[19:20].view(np.uint32)[0]) == 5 for b in blocks)
    a = 2 * math.pi * 7 / 120
    assert np.allclose(blocks[0][16:19], (3 * math.cos(a), 2.1, 3 * math.sin(a)))


def oracle_pose_renderer(osc, cfg, world, rank):
    """Stand-in for the device pass: the CPU oracle accumulates the pose's frames, the rank keeps its own rows."""
    def render_pose(blocks, color):
        rgba = np.zeros((cfg.height, cfg.width, 4), np.float32)
        for k, u in enumerate(blocks):
            rgba = osc.render(u, cfg.width, cfg.height, accum_index=k + 1, blend=True, rgba=rgba, taps=False, threads=2)["rgba"]
        if world == 1:
            color.copy_(torch.from_numpy(rgba))
            return
        color.fill_(-1.0)
        for lr, y in enumerate(rdist.local_to_frame_rows(cfg.height, cfg.band_rows, world, rank)):
            if y >= 0:
                color[lr] = torch.from_numpy(rgba[y])
    return render_pose


def expected_frames(osc, cfg):
    out = {}
    for pose in POSES:
        rgba = np.zeros((cfg.height, cfg.width, 4), np.float32)
        for k, u in enumerate(orbit.pose_uniforms(cfg, pose)):
            rgba = osc.render(u, cfg.width, cfg.height, accum_index=k + 1, blend=True, rgba=rgba, taps=False, threads=2)["rgba"]
        out[pose] = rgba
    return out


def _scene_and_oracle():
    from oracle import oracle
    sky = np.random.default_rng(1).integers(0, 256, size=(6, 16, 16, 4), dtype=np.uint8)
    return oracle.OracleScene(scenes.small_test_scene().as_dict(), sky)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = orbit.OrbitConfig(**CFG)
        osc = _scene_and_oracle()
        got = []
        drv = orbit.OrbitDriver(cfg, oracle_pose_renderer(osc, cfg, world, rank), torch.device("cpu"), world=world, rank=rank)
        drv.run(POSES, sink=lambda pose, frame: got.append((pose, frame.clone().numpy())))
        ok = True
        if rank == 0:
            want = expected_frames(osc, cfg)
            ok = [p for p, _ in got] == POSES and all(np.array_equal(f.view(np.uint32), want[p].view(np.uint32)) for p, f in got)
        else:
            ok = got == []
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_orbit_driver_two_ranks_equals_full_frames():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=300) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert res == {0: True, 1: True}


def test_orbit_driver_single_rank():
    cfg = orbit.OrbitConfig(**CFG)
    osc = _scene_and_oracle()
    got = []
    orbit.OrbitDriver(cfg, oracle_pose_renderer(osc, cfg, 1, 0), torch.device("cpu")).run(POSES, sink=lambda p, f: got.append((p, f.clone().numpy())))
    want = expected_frames(osc, cfg)
    assert [p for p, _ in got] == POSES
    assert all(np.array_equal(f.view(np.uint32), want[p].view(np.uint32)) for p, f in got)
    assert not np.array_equal(want[0], want[37])           # the poses really differ


def test_scene_pose_renderer_refuses_host_memory():
    import pytest
    cfg = orbit.OrbitConfig(**CFG)
    render_pose = orbit.scene_pose_renderer(object(), cfg)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        render_pose(orbit.pose_uniforms(cfg, 0), torch.zeros((cfg.height, cfg.width, 4)))
