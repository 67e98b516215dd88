"""GPU tests of the camera-path workload (7680x4320, accumulated frames per pose on a fixed orbit, interleaved bands):
the driver against the oracle at a small size, and properties at the full size."""
import numpy as np
import pytest
import torch

from raytracer_amd import api, dist as rdist, orbit

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def scene(bunny_octree, test_sky):
    sc = api.Scene.from_octree(bunny_octree)
    sc.set_skybox(test_sky)
    yield sc
    sc.close()


def test_orbit_poses_match_oracle(scene, bunny_octree, test_sky):
    from oracle import oracle
    cfg = orbit.OrbitConfig(width=160, height=96, spp=4)
    osc = oracle.OracleScene(bunny_octree.as_dict(), test_sky)
    poses = [0, 17, 45, 93]
    got = {}
    drv = orbit.OrbitDriver(cfg, orbit.scene_pose_renderer(scene, cfg), torch.device("cuda", 0))
    drv.run(poses, sink=lambda p, f: got.__setitem__(p, f.cpu().numpy()))
    assert sorted(got) == sorted(poses)
    for p in poses:
        rgba = np.zeros((cfg.height, cfg.width, 4), np.float32)
        for k, u in enumerate(orbit.pose_uniforms(cfg, p)):
            rgba = osc.render(u, cfg.width, cfg.height, accum_index=k + 1, rgba=rgba, taps=False)["rgba"]
        assert np.array_equal(bits(got[p]), bits(rgba)), p


def test_8k_band_share_of_a_pose(scene, bunny_octree, test_sky):
    """One of eight ranks' share of a 7680x4320 pose (3 accumulated frames, traced in one batched launch): sampled strips
    against the oracle, and the same rows of the full-frame render."""
    from oracle import oracle
    cfg = orbit.OrbitConfig(spp=3)
    W, H, parts, part = cfg.width, cfg.height, 8, 5
    blocks = orbit.pose_uniforms(cfg, 11)
    dev = torch.device("cuda", 0)
    rows = rdist.local_row_count(H, cfg.band_rows, parts, part)
    color = torch.zeros((rdist.padded_row_count(H, cfg.band_rows, parts), W, 4), dtype=torch.float32, device=dev)
    orbit.scene_pose_renderer(scene, cfg, world=parts, rank=part)(blocks, color)
    full = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    orbit.scene_pose_renderer(scene, cfg)(blocks, full)
    torch.cuda.synchronize()
    ys = rdist.local_to_frame_rows(H, cfg.band_rows, parts, part)
    assert (ys >= 0).all() and len(ys) == rows
    share, whole = color[:rows].cpu().numpy(), full.cpu().numpy()
    assert np.array_equal(bits(share), bits(whole[ys]))
    osc = oracle.OracleScene(bunny_octree.as_dict(), test_sky)
    for strip in (0, 12, 20, 33):                                  # local strips: ground, bunny, spheres, sky
        y0 = int(ys[strip * cfg.band_rows])
        rgba = np.zeros((H, W, 4), np.float32)
        for k, u in enumerate(blocks):
            rgba = osc.render(u, W, H, accum_index=k + 1, rgba=rgba, rows=(y0, y0 + cfg.band_rows), taps=False)["rgba"]
        assert np.array_equal(bits(share[strip * cfg.band_rows:(strip + 1) * cfg.band_rows]), bits(rgba[y0:y0 + cfg.band_rows])), strip
