"""GPU tests of rt_overlay_draw (include/rt_overlay.h): the HIP raster pass against llvmpipe runs of the reference's raster
shaders (fixtures), against the CPU restatement at larger sizes, and chained after the HIP ray tracer as in App::Render."""
import glob
import os

import numpy as np
import pytest

from oracle import raster
from raytracer_amd import api

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SUZANNE_FILES = sorted(glob.glob(os.path.join(GOLDEN, "ref_overlay_suzanne_*.npz")))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def split_words(words):
    """(view_proj, worlds, world_its, light, la, ld) of the fixture's uniform words."""
    n = (len(words) - 26) // 32
    inst = words[26:].reshape(n, 32)
    return words[:16], inst[:, :16], inst[:, 16:], words[16:19], words[19:22], words[22:25]


@pytest.mark.parametrize("k", range(8))
def test_triangle_soups_match_llvmpipe(k):
    z = np.load(os.path.join(GOLDEN, "ref_overlay_soups.npz"))
    h, w = z[f"depth24_{k}"].shape
    m = api.OverlayMesh(z[f"vertices_{k}"], z[f"indices_{k}"], z[f"texture_{k}"] if f"texture_{k}" in z.files else z["texture"])
    col = np.zeros((h, w, 4), np.float32); d24 = np.full((h, w), 0xFFFFFF, np.uint32)
    m.draw(col, d24, *split_words(z[f"words_{k}"]))
    m.close()
    assert np.array_equal(d24, z[f"depth24_{k}"])
    assert np.array_equal(bits(col), bits(z[f"rgba_{k}"]))


@pytest.mark.parametrize("path", SUZANNE_FILES, ids=[os.path.basename(p)[20:-4] for p in SUZANNE_FILES])
def test_reference_meshes_match_llvmpipe(path, bunny_octree):
    """The nine draws alone; over llvmpipe's ray-traced frame; and over the HIP ray tracer's own frame (rt_render ->
    rt_depth_to_attachment -> rt_overlay_draw), which must equal llvmpipe's finished frame of App::Render bit for bit."""
    a = np.load(os.path.join(GOLDEN, "overlay_assets.npz")); z = np.load(path)
    w, h = (int(v) for v in z["size"])
    m = api.OverlayMesh(a["vertices"], a["indices"], a["texture"])
    args = split_words(z["words"])
    col = np.zeros((h, w, 4), np.float32); d24 = np.full((h, w), 0xFFFFFF, np.uint32)
    m.draw(col, d24, *args)
    assert np.array_equal(d24, z["alone_depth24"]) and np.array_equal(bits(col), bits(z["alone_rgba"]))
    col = z["traced_rgba"].copy(); d24 = z["traced_depth24"].copy()
    m.draw(col, d24, *args)
    assert np.array_equal(d24, z["frame_depth24"]) and np.array_equal(bits(col), bits(z["frame_rgba"]))
    sc = api.Scene.from_octree(bunny_octree); sc.set_skybox(z["sky"])
    got = sc.render(z["uniforms"], w, h, planes=False)
    sc.close()
    col = np.ascontiguousarray(got["rgba"], np.float32); d24 = api.depth_to_attachment(got["depth"])
    assert np.array_equal(d24, z["traced_depth24"]) and np.array_equal(bits(col), bits(z["traced_rgba"]))
    m.draw(col, d24, *args)
    m.close()
    assert np.array_equal(d24, z["frame_depth24"]) and np.array_equal(bits(col), bits(z["frame_rgba"]))


def test_three_frames_of_app_render(bunny_octree):
    """rt_render (blend 1/k over the finished previous frame) -> rt_depth_to_attachment -> rt_overlay_draw, three times, on device
    planes: llvmpipe's third frame of the reference's whole Render() bit for bit."""
    import torch
    a = np.load(os.path.join(GOLDEN, "overlay_assets.npz")); z = np.load(os.path.join(GOLDEN, "ref_overlay_suzanne_through.npz"))
    w, h = (int(v) for v in z["size"])
    sc = api.Scene.from_octree(bunny_octree); sc.set_skybox(z["sky"])
    m = api.OverlayMesh(a["vertices"], a["indices"], a["texture"])
    col = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda"); depth = torch.zeros((h, w), dtype=torch.float32, device="cuda")
    d24 = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    rp = api.RenderPass(sc, w, h, col.data_ptr(), depth_ptr=depth.data_ptr(), stream=st)
    for t in z["times"]:
        u = z["uniforms"].copy(); u[20] = t
        rp.Render(u)
        api.depth_to_attachment(depth, out=d24)
        m.draw(col, d24, *split_words(z["words"]))
    shown = api.present_rgba8(col)                      # Framebuffer::Blit: what the window receives
    torch.cuda.synchronize()
    assert np.array_equal(d24.cpu().numpy().view(np.uint32), z["three_depth24"])
    assert np.array_equal(bits(col.cpu().numpy()), bits(z["three_rgba"]))
    assert np.array_equal(shown.cpu().numpy(), z["three_present"])
    m.close(); sc.close()


def test_presented_bytes_of_probe_colours():
    z = np.load(os.path.join(GOLDEN, "ref_present.npz"))
    assert np.array_equal(api.present_rgba8(z["color"]), z["rgba8"])


def test_4k_frame_on_device_equals_the_cpu_restatement():
    """The reference's nine draws at 3840 x 2160 on device-resident planes against oracle/raster_oracle.c."""
    import torch
    a = np.load(os.path.join(GOLDEN, "overlay_assets.npz"))
    from raytracer_amd.camera import Camera
    w, h = 3840, 2160
    cam = Camera(eye=(-2.5, 5.25, 11.0), at=(-2.5, 5.25, 4.1), aspect=w / h)
    worlds, its = raster.reference_instances()
    words = raster.overlay_words(cam.view_proj().ravel(), worlds, its, **raster.REFERENCE_LIGHT)
    rng = np.random.default_rng(3)
    col0 = rng.random((h, w, 4), dtype=np.float32); d0 = rng.integers(0xFF9800, 0x1000000, (h, w), dtype=np.uint32)      # around the meshes' depths: some of their fragments are hidden
    want_c, want_d = col0.copy(), d0.copy()
    raster.draw(a["vertices"], a["indices"], words, a["texture"], want_c, want_d)
    assert (want_d != d0).sum() > 500_000
    m = api.OverlayMesh(a["vertices"], a["indices"], a["texture"])
    col = torch.from_numpy(col0).cuda(); d24 = torch.from_numpy(d0.astype(np.int32)).cuda()
    for _ in range(2):      # the second call reuses the mesh's scratch; drawing twice over its own result changes nothing (LEQUAL, same fragments)
        m.draw(col, d24, *split_words(words))
    torch.cuda.synchronize()
    m.close()
    assert np.array_equal(d24.cpu().numpy().view(np.uint32), want_d)
    assert np.array_equal(bits(col.cpu().numpy()), bits(want_c))


def test_equal_depths_keep_the_last_draw_and_arguments_are_checked():
    v = np.zeros((6, 8), np.float32)
    v[:3, :3] = [[-0.5, -0.5, 0.25], [0.5, -0.5, 0.25], [0.0, 0.5, 0.25]]; v[3:, :3] = v[:3, :3]
    v[:3, 3:6] = (0, 0, 1); v[3:, 3:6] = (1, 2, 0.5)      # the second triangle faces the light, the first does not
    tex = np.full((2, 2, 4), 255, np.uint8)
    I = np.eye(4, dtype=np.float32).reshape(-1)
    col = np.zeros((32, 32, 4), np.float32); d24 = np.full((32, 32), 0xFFFFFF, np.uint32)
    m = api.OverlayMesh(v, np.arange(6, dtype=np.uint32), tex)
    m.draw(col, d24, I, [I], [I], (-1, -2, -0.5), (0, 0, 0), (1, 1, 1))
    want_c = np.zeros_like(col); want_d = np.full_like(d24, 0xFFFFFF)
    raster.draw(v, np.arange(6, dtype=np.uint32), raster.overlay_words(I, [I], [I], (-1, -2, -0.5), (0, 0, 0), (1, 1, 1)), tex, want_c, want_d)
    assert np.array_equal(d24, want_d) and np.array_equal(bits(col), bits(want_c))
    assert col[16, 16, 0] > 0.9      # lit: the later triangle's colour
    with pytest.raises(api.RtError):
        api.OverlayMesh(v, np.array([0, 1, 7], np.uint32), tex)
    with pytest.raises(api.RtError):
        api.OverlayMesh(v, np.array([0, 1], np.uint32), tex)
    m.close()


def test_random_soups_equal_the_cpu_restatement():
    """Random soups (non-power-of-two frames and textures, perspective, several draws, slivers, clipping) on the GPU against
    oracle/raster_oracle.c, which the by-hand run `python -m oracle.fuzz_raster_vs_reference` holds to llvmpipe."""
    from raytracer_amd.camera import Camera
    rng = np.random.default_rng(11)
    F = np.float32
    I = np.eye(4, dtype=F)
    for k in range(24):
        w, h = [(96, 64), (100, 75), (121, 67), (333, 190), (64, 128)][rng.integers(0, 5)]
        n = int(rng.integers(5, 80))
        v = np.zeros((3 * n, 8), F)
        spread, size = [(0.6, 0.4), (1.5, 1.2), (0.9, 0.05), (3.0, 3.0)][rng.integers(0, 4)]
        c = rng.uniform(-spread, spread, (n, 1, 3)); v[:, :3] = (c + rng.uniform(-size, size, (n, 3, 3))).reshape(-1, 3)
        if k % 5 == 0:
            v[2::3, :3] = v[1::3, :3] + rng.normal(0, 1e-3, (n, 3))
        v[:, 3:6] = rng.uniform(-1, 1, (3 * n, 3))
        v[:, 6:8] = rng.uniform(-3, 3, (3 * n, 2)) * float(10 ** rng.uniform(-2, 1)) if k % 3 else rng.uniform(-3, 3, 2)      # constant, magnified or minified
        th, tw = [(4, 4), (16, 8), (7, 5), (32, 32), (8, 5), (6, 16), (1, 1), (3, 64)][rng.integers(0, 8)]
        tex = rng.integers(0, 256, (th, tw, 4), dtype=np.uint8)
        if k % 2:
            e = rng.normal(size=3); e = e / np.linalg.norm(e) * rng.uniform(0.5, 4.0)
            vp = Camera(eye=tuple(float(x) for x in e), at=tuple(rng.uniform(-0.3, 0.3, 3)), fovy=float(rng.uniform(0.4, 2.0)), aspect=w / h,
                        z_near=float(10 ** rng.uniform(-2, -0.2)), z_far=float(10 ** rng.uniform(0.5, 2))).view_proj().ravel()
        else:
            vp = I.reshape(-1)
        worlds, its = [], []
        for _ in range(int(rng.integers(1, 4))):
            m = np.eye(4); m[:3, :3] = np.linalg.qr(rng.normal(size=(3, 3)))[0] * rng.uniform(0.5, 1.5); m[:3, 3] = rng.uniform(-0.5, 0.5, 3)
            worlds.append(m.T.astype(F).reshape(-1)); its.append(np.linalg.inv(m).astype(F).reshape(-1))
        light, la, ld = rng.uniform(-1, 1, 3), rng.uniform(0, 0.5, 3), rng.uniform(0, 1, 3)
        idx = np.arange(3 * n, dtype=np.uint32)
        want_c = rng.random((h, w, 4), dtype=np.float32); want_d = rng.integers(0x800000, 0x1000000, (h, w), dtype=np.uint32)
        col, d24 = want_c.copy(), want_d.copy()
        raster.draw(v, idx, raster.overlay_words(vp, worlds, its, light, la, ld), tex, want_c, want_d)
        m = api.OverlayMesh(v, idx, tex)
        m.draw(col, d24, vp, worlds, its, light, la, ld)
        m.close()
        assert np.array_equal(d24, want_d), k
        assert np.array_equal(bits(col), bits(want_c)), k


@pytest.mark.parametrize("band_rows,parts", [(16, 2), (5, 3), (16, 8)])
def test_band_draws_reassemble_the_frame(band_rows, parts):
    """rt_overlay_draw_band: every owner draws its strips (full-frame planes left alone elsewhere, or compact planes of its rows only);
    together they give the whole-frame draw bit for bit, mip-mapped lookups (quads span two rows) included."""
    a = np.load(os.path.join(GOLDEN, "overlay_assets.npz"))
    from raytracer_amd.camera import Camera
    w, h = 200, 150
    rng = np.random.default_rng(band_rows * 10 + parts)
    v = a["vertices"].copy(); v[:, 6:8] = rng.uniform(-4, 4, (len(v), 2))      # give the mesh texture coordinates: minified lookups
    tex = rng.integers(0, 256, (24, 40, 4), dtype=np.uint8)
    cam = Camera(eye=(-2.5, 5.25, 9.0), at=(-2.5, 5.25, 4.1), aspect=w / h)
    worlds, its = raster.reference_instances()
    args = (cam.view_proj().ravel(), worlds, its, (-1.0, -2.0, -0.5), (0.1, 0.1, 0.1), (1.0, 1.0, 1.0))
    col0 = rng.random((h, w, 4), dtype=np.float32); d0 = rng.integers(0xFF9800, 0x1000000, (h, w), dtype=np.uint32)
    m = api.OverlayMesh(v, a["indices"], tex)
    want_c, want_d = col0.copy(), d0.copy()
    m.draw(want_c, want_d, *args)
    assert (want_d != d0).sum() > 2000
    full_c, full_d = col0.copy(), d0.copy()
    comp_c, comp_d = col0.copy(), d0.copy()
    for part in range(parts):
        m.draw(full_c, full_d, *args, band=(band_rows, parts, part, False))
        rows = np.array([y for y in range(h) if (y // band_rows) % parts == part])
        if len(rows) == 0:
            continue
        pc, pd = np.ascontiguousarray(col0[rows]), np.ascontiguousarray(d0[rows])
        m.draw(pc, pd, *args, band=(band_rows, parts, part, True), frame_height=h)
        comp_c[rows], comp_d[rows] = pc, pd
    m.close()
    assert np.array_equal(full_d, want_d) and np.array_equal(bits(full_c), bits(want_c))
    assert np.array_equal(comp_d, want_d) and np.array_equal(bits(comp_c), bits(want_c))
