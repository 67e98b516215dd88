"""CPU tests: the restatement of the raster pass (oracle/raster_oracle.c) against llvmpipe runs of the reference's
Vert_PosNormTex.vert / Frag_LightingSimple.frag with the state of App.cpp:46-56,142-166 (fixtures from oracle/gen_fixtures.py overlay)."""
import glob
import os

import numpy as np
import pytest

from oracle import raster

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SUZANNE_FILES = sorted(glob.glob(os.path.join(GOLDEN, "ref_overlay_suzanne_*.npz")))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("k", range(8))
def test_triangle_soups_bit_exact(k):
    """Clipped by every frustum plane (2-5), perspective (odd k), shuffled draw order (4), edges through pixel centres (5), minified
    lookups through all mip levels (6, 7)."""
    z = np.load(os.path.join(GOLDEN, "ref_overlay_soups.npz"))
    h, w = z[f"depth24_{k}"].shape
    col = np.zeros((h, w, 4), np.float32); d24 = np.full((h, w), 0xFFFFFF, np.uint32)
    raster.draw(z[f"vertices_{k}"], z[f"indices_{k}"], z[f"words_{k}"], z[f"texture_{k}"] if f"texture_{k}" in z.files else z["texture"], col, d24)
    assert np.array_equal(d24, z[f"depth24_{k}"])
    assert np.array_equal(bits(col), bits(z[f"rgba_{k}"]))
    assert (d24 != 0xFFFFFF).sum() > 500


@pytest.mark.parametrize("path", SUZANNE_FILES, ids=[os.path.basename(p)[20:-4] for p in SUZANNE_FILES])
def test_reference_meshes_alone_and_in_the_frame(path):
    """The nine draws of App.cpp:157-164 over an empty frame, and over the ray-traced frame llvmpipe produced (its colour and
    24-bit depth are the inputs): both must reproduce llvmpipe's read-backs of the finished frame bit for bit."""
    a = np.load(os.path.join(GOLDEN, "overlay_assets.npz")); z = np.load(path)
    w, h = (int(v) for v in z["size"])
    col = np.zeros((h, w, 4), np.float32); d24 = np.full((h, w), 0xFFFFFF, np.uint32)
    raster.draw(a["vertices"], a["indices"], z["words"], a["texture"], col, d24)
    assert np.array_equal(d24, z["alone_depth24"]) and np.array_equal(bits(col), bits(z["alone_rgba"]))
    col = z["traced_rgba"].copy(); d24 = z["traced_depth24"].copy()
    raster.draw(a["vertices"], a["indices"], z["words"], a["texture"], col, d24)
    assert np.array_equal(d24, z["frame_depth24"]) and np.array_equal(bits(col), bits(z["frame_rgba"]))
    assert (z["frame_depth24"] != z["traced_depth24"]).sum() > 1000


def test_three_frames_of_the_whole_pass():
    """App::Render three times with the camera at rest: the ray-tracing draw of frame k blends 1/k over the finished frame k-1
    (rasterised mesh pixels included: the colour attachment persists, App.cpp:96-105), then the raster pass runs again."""
    from oracle import oracle
    from oracle.gen_fixtures import bunny_scene
    a = np.load(os.path.join(GOLDEN, "overlay_assets.npz")); z = np.load(os.path.join(GOLDEN, "ref_overlay_suzanne_through.npz"))
    w, h = (int(v) for v in z["size"])
    osc = oracle.OracleScene(bunny_scene(), z["sky"])
    col = None
    for k, t in enumerate(z["times"]):
        u = z["uniforms"].copy(); u[20] = t
        o = osc.render(u, w, h, accum_index=k + 1, rgba=col, taps=False)
        col = np.ascontiguousarray(o["rgba"], np.float32)
        d24 = np.rint((np.clip(o["depth"], 0, 1) * np.float32(16777215.0)).astype(np.float32)).astype(np.uint32)
        raster.draw(a["vertices"], a["indices"], z["words"], a["texture"], col, d24)
    assert np.array_equal(d24, z["three_depth24"]) and np.array_equal(bits(col), bits(z["three_rgba"]))


def test_presented_bytes():
    """Framebuffer::Blit onto the window's 8-bit framebuffer (Framebuffer.cpp:33-35): the probe colours and the third frame of the
    whole Render() of the fixture above."""
    from oracle import oracle
    z = np.load(os.path.join(GOLDEN, "ref_present.npz"))
    assert np.array_equal(oracle.present_rgba8(z["color"]), z["rgba8"])
    t = np.load(os.path.join(GOLDEN, "ref_overlay_suzanne_through.npz"))
    assert np.array_equal(oracle.present_rgba8(t["three_rgba"]), t["three_present"])
    assert t["three_present"][..., 3].max() == 255 and t["three_present"][..., 3].min() == 0      # alpha: 1 on mesh pixels, 0 elsewhere (frag:670)


def test_instance_matrices_and_light_of_the_reference():
    worlds, its = raster.reference_instances()
    assert len(worlds) == 9 and worlds[0][12:15].tolist() == [-4.0, 0.0, 0.0] and worlds[8][12:15].tolist() == [4.0, 8.0, 0.0]      # App.cpp:157-159
    for w, it in zip(worlds, its):
        m = w.reshape(4, 4).T.astype(np.float64)
        assert np.allclose(np.linalg.inv(m).T, it.reshape(4, 4).T)


def test_mesh_loader_shapes():
    a = np.load(os.path.join(GOLDEN, "overlay_assets.npz"))
    assert a["vertices"].shape == (507, 8) and a["indices"].shape == (2904,) and a["indices"].max() == 506      # 32 triangles + 468 quads
    assert not a["vertices"][:, 6:8].any()      # the mesh file has no texture coordinates: every vertex gets (0, 0) (ObjParser.cpp:406)
