"""GPU tests of include/rt_overlay.h through the C ABI: the 24-bit depth-attachment values against the read-backs of
the reference runs on llvmpipe (tests/golden), and the depth-tested merge of rasterised fragments against the CPU
restatement (GL_LEQUAL, depth writes, no blending: /root/reference/app/core/src/App.cpp:54,142-145)."""
import glob
import os

import numpy as np
import pytest
import torch

from raytracer_amd import api

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ATTACHMENT_FILES = sorted(f for f in glob.glob(os.path.join(GOLDEN, "ref*.npz")) if "attachment_depth" in np.load(f).files)


def attachment_values(z):
    return np.rint(z["attachment_depth"].astype(np.float64) * 16777215.0).astype(np.uint32)


@pytest.mark.parametrize("path", ATTACHMENT_FILES, ids=[os.path.basename(p)[:-4] for p in ATTACHMENT_FILES])
def test_attachment_values_match_llvmpipe(path):
    z = np.load(path)
    want = attachment_values(z)
    assert np.array_equal(api.depth_to_attachment(z["depth"]), want)                                     # host planes (staged)
    d = torch.from_numpy(np.ascontiguousarray(z["depth"])).cuda()
    got = api.depth_to_attachment(d)
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy().view(np.uint32), want)                                       # device planes


def test_attachment_of_a_rendered_frame(bunny_octree, test_sky):
    """The whole chain on the device: render into a device depth plane, convert, compare with the reference's attachment."""
    z = np.load(os.path.join(GOLDEN, "ref_default.npz"))
    h, w = z["depth"].shape
    sc = api.Scene.from_octree(bunny_octree)
    sc.set_skybox(z["sky"])
    r = sc.render(z["uniforms"], w, h)
    sc.close()
    assert np.array_equal(api.depth_to_attachment(r["depth"]), attachment_values(z))


def test_attachment_edge_values_and_ragged_sizes():
    from oracle import oracle
    edge = np.array([-1.0, -0.0, 0.0, 1e-45, 2.9802322e-08, 0.25, 0.5, 0.99999994, 1.0, 1.5, np.inf, -np.inf, np.nan], np.float32)
    assert np.array_equal(api.depth_to_attachment(edge), oracle.depth_attachment(edge))
    assert api.depth_to_attachment(np.zeros(0, np.float32)).size == 0
    rng = np.random.default_rng(3)
    for n in (1, 3, 4, 5, 255, 1027, 65537):
        d = rng.uniform(-0.1, 1.1, n).astype(np.float32)
        assert np.array_equal(api.depth_to_attachment(d), oracle.depth_attachment(d)), n
        t = torch.from_numpy(np.concatenate([np.zeros(1, np.float32), d])).cuda()[1:]                    # a plane that is not 16-byte aligned
        got = api.depth_to_attachment(t.contiguous() if n % 2 else t)
        torch.cuda.synchronize()
        assert np.array_equal(got.cpu().numpy().view(np.uint32), oracle.depth_attachment(d)), n


def layer(rng, n, p_cover=0.6):
    return (rng.uniform(0, 1, (n, 4)).astype(np.float32), rng.integers(0, 1 << 24, n, dtype=np.uint32), (rng.uniform(0, 1, n) < p_cover).astype(np.uint8))


def test_composite_matches_the_restatement():
    from oracle import oracle
    rng = np.random.default_rng(11)
    for n in (1, 63, 64 * 48, 100003):
        color = rng.uniform(0, 1, (n, 4)).astype(np.float32)
        z = rng.integers(0, 1 << 24, n, dtype=np.uint32)
        fc, fz, mask = layer(rng, n)
        fz[::3] = z[::3]                                             # equal depths pass (LEQUAL)
        want_c, want_z = oracle.overlay_composite(color, z, fc, fz, mask)
        got_c, got_z = color.copy(), z.copy()
        api.overlay_composite(got_c, got_z, fc, fz, mask)
        assert np.array_equal(got_c.view(np.uint32), want_c.view(np.uint32)) and np.array_equal(got_z, want_z)
        assert (want_z <= z).all() and (want_z[::3][mask[::3] != 0] == z[::3][mask[::3] != 0]).all()
        # no mask: every pixel carries a fragment
        want_c, want_z = oracle.overlay_composite(color, z, fc, fz, None)
        got_c, got_z = color.copy(), z.copy()
        api.overlay_composite(got_c, got_z, fc, fz)
        assert np.array_equal(got_c.view(np.uint32), want_c.view(np.uint32)) and np.array_equal(got_z, want_z)


def test_composite_on_device_planes_over_a_4k_frame():
    """Full-size properties on device memory: nine layers applied in draw order equal the CPU restatement's result, the
    result keeps the nearest fragment of every pixel, and applying a layer twice changes nothing."""
    from oracle import oracle
    rng = np.random.default_rng(12)
    w, h = 3840, 2160
    n = w * h
    color = np.zeros((n, 4), np.float32)
    depth = rng.uniform(0.9, 1.0, n).astype(np.float32)
    z = api.depth_to_attachment(depth)
    d_color, d_z = torch.from_numpy(color).cuda(), torch.from_numpy(z.view(np.int32)).cuda()
    want_c, want_z = color, z
    layers = []
    for k in range(9):
        fc = np.full((n, 4), k + 1, np.float32)
        fz = rng.integers(int(0.85 * (1 << 24)), 1 << 24, n, dtype=np.uint32)
        mask = np.zeros(n, np.uint8)
        lo = rng.integers(0, n - n // 6)
        mask[lo:lo + n // 6] = 1
        layers.append((fc, fz, mask))
        want_c, want_z = oracle.overlay_composite(want_c, want_z, fc, fz, mask)
        api.overlay_composite(d_color, d_z, torch.from_numpy(fc).cuda(), torch.from_numpy(fz.view(np.int32)).cuda(), torch.from_numpy(mask).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(d_color.cpu().numpy(), want_c) and np.array_equal(d_z.cpu().numpy().view(np.uint32), want_z)
    nearest = z.copy()
    for fc, fz, mask in layers:
        nearest = np.where(mask != 0, np.minimum(nearest, fz), nearest)
    assert np.array_equal(want_z, nearest)
    fc, fz, mask = layers[4]
    api.overlay_composite(d_color, d_z, torch.from_numpy(fc).cuda(), torch.from_numpy(fz.view(np.int32)).cuda(), torch.from_numpy(mask).cuda())
    torch.cuda.synchronize()
    again = d_z.cpu().numpy().view(np.uint32)
    assert np.array_equal(again, want_z)                                    # depth cannot change: nothing is nearer than itself
    changed = (d_color.cpu().numpy() != want_c).any(-1)
    assert not changed[(mask == 0) | (fz > want_z)].any()                   # only fragments that tie with the stored depth may repaint


def test_argument_errors():
    L = api.lib()
    z = np.zeros(4, np.uint32); d = np.zeros(4, np.float32)
    assert L.rt_depth_to_attachment(None, z.ctypes.data, 4, api.RT_MEM_HOST, None) == api.RT_EINVAL
    assert L.rt_depth_to_attachment(d.ctypes.data, z.ctypes.data, 4, 7, None) == api.RT_EINVAL
    assert L.rt_overlay_composite(None, None, None, None, None, 4, api.RT_MEM_HOST, None) == api.RT_EINVAL
    assert L.rt_overlay_composite(None, None, None, None, None, 0, api.RT_MEM_HOST, None) == api.RT_OK
