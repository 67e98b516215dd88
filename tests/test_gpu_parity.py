"""GPU parity: the HIP render pass (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar: bit-exact primary-hit kind / slot / t / segment counts; colour bit-exact (the stated tolerance of
max-abs <= 1/255 on >= 99.9 % of pixels is asserted as the fallback bar and PSNR is reported)."""
import math

import numpy as np
import pytest

from raytracer_amd import api, scenes
from raytracer_amd.camera import Camera
from raytracer_amd.params import Uniforms

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def compare(o, g, name, exact=True):
    rep = {}
    for k in ("kind", "slot", "nrays"):
        rep[k] = float((o[k] == g[k]).mean())
    rep["t"] = float((bits(o["t"]) == bits(g["t"])).mean())
    rep["depth"] = float((bits(o["depth"]) == bits(g["depth"])).mean())
    ceq = (bits(o["rgba"]) == bits(g["rgba"])).all(-1)
    rep["colour_bits"] = float(ceq.mean())
    d = np.abs(o["rgba"][..., :3].astype(np.float64) - g["rgba"][..., :3].astype(np.float64))
    d = np.where(np.isnan(d), np.inf, d)
    rep["within_1_255"] = float((d.max(-1) <= 1.0 / 255.0).mean())
    mse = float(np.mean(np.minimum(d, 1.0) ** 2))
    rep["psnr_db"] = float("inf") if mse == 0 else 10.0 * math.log10(1.0 / mse)
    print(name, rep)
    assert rep["kind"] == 1.0 and rep["slot"] == 1.0 and rep["t"] == 1.0, (name, rep)   # hit IDs: bit-exact
    assert rep["nrays"] == 1.0, (name, rep)
    assert rep["within_1_255"] >= 0.999, (name, rep)                                     # colour tolerance bar
    if exact:
        assert rep["colour_bits"] == 1.0 and rep["depth"] == 1.0, (name, rep)
    return rep


VIEWS = {
    "default": dict(cam=Camera(aspect=1.0)),
    "wide": dict(cam=Camera(aspect=2.0), size=(512, 256)),
    "portal": dict(cam=Camera(eye=(3.0, 2.0, 3.0), at=(7.0, 2.0, 3.0), aspect=1.0)),
    "portal2": dict(cam=Camera(eye=(9.5, 2.2, 6.5), at=(7.0, 1.5, 4.0), aspect=1.0)),
    "glass": dict(cam=Camera(eye=(-2.0, 2.5, 3.0), at=(-4.0, 2.0, 0.0), aspect=1.0)),
    "close": dict(cam=Camera(eye=(0.5, 2.5, 1.0), at=(-0.3, 1.9, 0.0), aspect=1.0)),
    "time": dict(cam=Camera(aspect=1.0), kw=dict(time=3.7)),
    "topdown": dict(cam=Camera(eye=(0.1, 9.0, 0.1), at=(0.0, 0.0, 0.0), aspect=1.0)),
    "maxrec2": dict(cam=Camera(aspect=1.0), kw=dict(max_recursion_limit=2)),
    "maxrec8": dict(cam=Camera(aspect=1.0), kw=dict(max_recursion_limit=8, blur_amount=0.01)),
}


@pytest.fixture(scope="module", params=["wavefront", "megakernel", "persistent"])
def bunny_pair(request, bunny_octree, test_sky):
    from oracle import oracle
    sc = api.Scene.from_octree(bunny_octree)
    sc.set_skybox(test_sky)
    sc.set_pipeline(request.param)
    osc = oracle.OracleScene(bunny_octree.as_dict(), test_sky)
    yield sc, osc
    sc.close()


@pytest.mark.parametrize("view", list(VIEWS))
def test_bunny_views_bit_exact(bunny_pair, view):
    sc, osc = bunny_pair
    v = VIEWS[view]
    w, h = v.get("size", (256, 256))
    u = Uniforms.from_camera(v["cam"], **v.get("kw", {})).pack()
    compare(osc.render(u, w, h), sc.render(u, w, h), view)


def test_progressive_accumulation(bunny_pair):
    sc, osc = bunny_pair
    w = h = 128
    acc_o = acc_g = None
    for k, t in enumerate(np.array([0.0, 1 / 60, 2 / 60, 0.5], np.float32)):
        u = Uniforms.from_camera(Camera(aspect=1.0), time=float(t)).pack()
        acc_o = osc.render(u, w, h, accum_index=k + 1, rgba=acc_o)["rgba"]
        acc_g = sc.render(u, w, h, accum_index=k + 1, color=acc_g)["rgba"]
    assert np.array_equal(bits(acc_o), bits(acc_g))


def test_bands_equal_full_frame(bunny_pair):
    sc, _ = bunny_pair
    w, h = 256, 200   # height not a multiple of the strip
    u = Uniforms.from_camera(Camera(aspect=w / h)).pack()
    full = sc.render(u, w, h)
    for parts in (2, 3, 8):
        col = np.zeros((h, w, 4), np.float32)
        planes = None
        for p in range(parts):
            r = sc.render(u, w, h, color=col, band=(16, parts, p), planes=False)
            col = r["rgba"]
        assert np.array_equal(bits(col), bits(full["rgba"])), parts


def test_no_mesh_scene(test_sky):
    from oracle import oracle
    d = dict(vertices=np.zeros((0, 4), np.float32), normals=np.zeros((0, 4), np.float32), tri_refs=np.zeros((0, 4), np.uint32), nodes=np.zeros((0,), np.uint32),
             bounds_min=np.zeros(3, np.float32), bounds_max=np.zeros(3, np.float32))
    sc = api.Scene.from_dict(d)
    sc.set_skybox(test_sky)
    u = Uniforms.from_camera(Camera(aspect=1.0)).pack()
    d1 = dict(d); d1["vertices"] = np.zeros((1, 4), np.float32); d1["normals"] = np.zeros((1, 4), np.float32); d1["tri_refs"] = np.zeros((1, 4), np.uint32)
    compare(oracle.OracleScene(d1, test_sky).render(u, 128, 128), sc.render(u, 128, 128), "no-mesh")
    sc.close()


def test_synthetic_scene_all_materials(test_sky):
    from oracle import oracle
    oc = scenes.small_test_scene()
    sc = api.Scene.from_octree(oc)
    sc.set_skybox(test_sky)
    osc = oracle.OracleScene(oc.as_dict(), test_sky)
    for cam in (Camera(eye=(2.5, 2.5, 2.5), at=(0.0, 1.5, 0.0), aspect=1.0), Camera(eye=(0.0, 1.5, 0.2), at=(1.0, 1.6, 0.0), aspect=1.0)):
        u = Uniforms.from_camera(cam).pack()
        compare(osc.render(u, 256, 256), sc.render(u, 256, 256), "synthetic")
    sc.close()


def test_counters_match_oracle(bunny_pair):
    sc, osc = bunny_pair
    u = Uniforms.from_camera(Camera(aspect=1.0)).pack()
    sc.set_counting(True)
    sc.render(u, 256, 256)
    st = sc.stats()
    sc.set_counting(False)
    oc = osc.render(u, 256, 256)["counters"]
    for k in ("rays", "node_pops", "triangle_tests", "child_box_tests", "triangle_hits_shaded", "rays_entering_octree"):
        assert st[k] == oc[k], (k, st, oc)


def test_subframes_and_pipelines_identical_large_frame(bunny_octree, test_sky):
    """A frame big enough to be cut into sub-frames on internal streams: every execution strategy gives the same bits,
    in full-frame and in compact band addressing."""
    sc = api.Scene.from_octree(bunny_octree)
    sc.set_skybox(test_sky)
    w, h = 2048, 1080
    u = Uniforms.from_camera(Camera(aspect=w / h), max_recursion_limit=3).pack()
    sc.set_pipeline("megakernel")
    ref = sc.render(u, w, h)
    for pipe, n in [("wavefront", n) for n in (1, 2, 4, 7)] + [("persistent", n) for n in (1, 3)]:
        sc.set_pipeline(pipe)
        sc.set_subframes(n)
        g = sc.render(u, w, h)
        for k in ("rgba", "depth", "t"):
            assert np.array_equal(bits(g[k]), bits(ref[k])), (pipe, n, k)
        for k in ("kind", "slot", "nrays"):
            assert np.array_equal(g[k], ref[k]), (pipe, n, k)
    sc.set_pipeline("wavefront")
    # compact band buffers of 2 parts re-interleave to the same frame
    from raytracer_amd import dist as rdist
    sc.set_subframes(4)
    frame = np.zeros((h, w, 4), np.float32)
    for part in range(2):
        fr = api.make_frame(u, w, h, band=(16, 2, part), compact=True)
        rows = api.band_row_count(fr)
        buf = np.zeros((rows, w, 4), np.float32)
        tg = api.rt_target()
        tg.memory, tg.sync, tg.color = api.RT_MEM_HOST, 1, buf.ctypes.data
        sc.render_raw(fr, tg)
        ys = rdist.local_to_frame_rows(h, 16, 2, part)
        frame[ys[ys >= 0]] = buf[ys >= 0]
    assert np.array_equal(bits(frame), bits(ref["rgba"]))
    sc.close()


def test_render_sequence_equals_frame_by_frame(bunny_octree, test_sky):
    """rt_render_sequence (sub-frame chains pipelined across frames) accumulates to the same bits as one rt_render
    per frame, for every pipeline / sub-frame count, with per-frame uniforms (time advances) and a host target."""
    sc = api.Scene.from_octree(bunny_octree)
    sc.set_skybox(test_sky)
    w, h = 2048, 1080
    n = 4
    us = [Uniforms.from_camera(Camera(aspect=w / h), time=k / 60.0, max_recursion_limit=2 + (k & 1)).pack() for k in range(n)]
    sc.set_pipeline("megakernel")
    ref = np.zeros((h, w, 4), np.float32)
    for k in range(n):
        ref = sc.render(us[k], w, h, accum_index=k + 1, color=ref, planes=False)["rgba"]
    for pipe, sub in (("megakernel", 1), ("megakernel", 3), ("auto", 3), ("wavefront", 1), ("wavefront", 2), ("wavefront", 4), ("persistent", 1), ("persistent", 3)):
        sc.set_pipeline(pipe)
        sc.set_subframes(sub)
        buf = np.zeros((h, w, 4), np.float32)
        depth = np.zeros((h, w), np.float32)
        tg = api.rt_target()
        tg.memory, tg.sync, tg.color, tg.depth = api.RT_MEM_HOST, 1, buf.ctypes.data, depth.ctypes.data
        sc.render_sequence_raw([api.make_frame(us[k], w, h, accum_index=k + 1) for k in range(n)], tg)
        assert np.array_equal(bits(buf), bits(ref)), (pipe, sub)
    # frames of a sequence must agree in geometry
    with pytest.raises(api.RtError):
        sc.render_sequence_raw([api.make_frame(us[0], w, h), api.make_frame(us[1], w, h - 16, accum_index=2)], tg)
    sc.close()


def _sphere_field_poses():
    """Camera poses chosen to stress the culling of the sphere loop: inside / on the faces of the slab that holds the
    small spheres, axis-parallel and grazing views, eyes on cell boundaries of the sphere grid, eyes near and beyond
    the distance up to which culling is used, plus seeded random poses."""
    poses = [
        ((0.0, 0.2, 0.0), (5.0, 0.2, 0.0)), ((0.0, 0.2, 0.0), (0.0, 0.2, -5.0)), ((1.0, 0.45, 1.0), (8.0, 0.45, 1.0)),
        ((1.0, -0.05, 1.0), (8.0, 0.3, 2.0)), ((-8.245381, 0.2, -7.0), (9.0, 0.2, -7.0)), ((-6.245381, 0.3, -9.0), (-6.245381, 0.3, 9.0)),
        ((-12.0, 0.25, -12.0), (12.0, 0.25, 12.0)), ((0.3, 6.0, 0.3), (0.3, 0.0, 0.3001)), ((31.9, 3.0, 0.0), (0.0, 0.0, 0.0)),
        ((32.1, 3.0, 0.0), (0.0, 0.0, 0.0)), ((200.0, 40.0, 150.0), (0.0, 0.0, 0.0)), ((2.0, 0.1, 2.0), (2.5, 3.0, 2.2)),
        ((-4.0, 2.0, 3.5), (-4.0, 2.0, 0.0)), ((0.0, 30.0, 0.0), (0.01, 0.0, 0.0)), ((9.5, 0.21, 9.5), (-9.0, 0.19, -9.0)),
    ]
    rng = np.random.default_rng(7)
    for _ in range(12):
        eye = rng.uniform(-11.0, 11.0, 3); eye[1] = rng.uniform(-0.2, 6.0)
        at = rng.uniform(-8.0, 8.0, 3); at[1] = rng.uniform(0.0, 2.0)
        poses.append((tuple(float(x) for x in eye), tuple(float(x) for x in at)))
    return poses


@pytest.mark.parametrize("pipeline", ["wavefront", "megakernel", "persistent"])
def test_sphere_field_poses_bit_exact(test_sky, pipeline):
    from oracle import oracle
    d = dict(vertices=np.zeros((1, 4), np.float32), normals=np.zeros((1, 4), np.float32), tri_refs=np.zeros((1, 4), np.uint32), nodes=np.zeros((0,), np.uint32),
             bounds_min=np.zeros(3, np.float32), bounds_max=np.zeros(3, np.float32))
    sc = api.Scene.from_dict(d)
    sc.set_skybox(test_sky)
    sc.set_pipeline(pipeline)
    osc = oracle.OracleScene(d, test_sky)
    w, h = 192, 128
    for i, (eye, at) in enumerate(_sphere_field_poses()):
        u = Uniforms.from_camera(Camera(eye=eye, at=at, aspect=w / h), time=0.25 * i, max_recursion_limit=6, blur_amount=1e-5 if i % 3 else 0.02).pack()
        compare(osc.render(u, w, h), sc.render(u, w, h), "pose %d %s" % (i, (eye, at)))
    sc.close()


def test_auto_pipeline_selection_is_invisible(bunny_octree, test_sky):
    """Under the default automatic selection the library tries every pipeline on its first frames and keeps re-measuring;
    whichever runs, the frame is the same, and afterwards every pipeline has a measurement."""
    sc = api.Scene.from_octree(bunny_octree)
    sc.set_skybox(test_sky)
    w, h = 512, 384
    u = Uniforms.from_camera(Camera(aspect=w / h), max_recursion_limit=3).pack()
    sc.set_pipeline("megakernel")
    ref = sc.render(u, w, h)
    sc.set_pipeline("auto")
    for _ in range(10):
        g = sc.render(u, w, h)
        for k in ("rgba", "depth", "t"):
            assert np.array_equal(bits(g[k]), bits(ref[k])), k
        for k in ("kind", "slot", "nrays"):
            assert np.array_equal(g[k], ref[k]), k
    selected, ms = sc.pipeline_state()
    assert all(v is not None and v > 0 for v in ms.values()), ms
    assert selected == min(ms, key=ms.get)
    sc.close()


def test_batched_sequence_of_small_frames(bunny_octree, test_sky):
    """Small launches inside rt_render_sequence are traced several frames at a time (samples staged, then blended in
    frame order): colour, depth and the debug planes equal frame-by-frame rendering, also on a band part in compact
    addressing, with a frame that stores instead of blending and one with no path segments in the middle."""
    sc = api.Scene.from_octree(bunny_octree)
    sc.set_skybox(test_sky)
    sc.set_pipeline("megakernel")
    w, h, n = 640, 360, 11
    us = [Uniforms.from_camera(Camera(aspect=w / h), time=k / 60.0, max_recursion_limit=(0 if k == 6 else 3)).pack() for k in range(n)]
    blends = [k != 4 for k in range(n)]
    for band, compact in (((16, 1, 0), False), ((16, 3, 1), True), ((16, 2, 0), False)):
        fr0 = api.make_frame(us[0], w, h, band=band, compact=compact)
        rows = api.band_row_count(fr0) if compact else h
        def fresh():
            return dict(color=np.full((rows, w, 4), 0.25, np.float32), depth=np.zeros((rows, w), np.float32), kind=np.zeros((rows, w), np.uint32),
                        slot=np.zeros((rows, w), np.uint32), t=np.zeros((rows, w), np.float32), nrays=np.zeros((rows, w), np.uint32))
        def target(b):
            tg = api.rt_target()
            tg.memory, tg.sync = api.RT_MEM_HOST, 1
            tg.color, tg.depth, tg.hit_kind, tg.hit_slot, tg.hit_t, tg.n_rays = (b[k].ctypes.data for k in ("color", "depth", "kind", "slot", "t", "nrays"))
            return tg
        frames = [api.make_frame(us[k], w, h, accum_index=k + 1, blend=blends[k], band=band, compact=compact) for k in range(n)]
        ref = fresh()
        sc.set_sequence_batch(1)
        for fr in frames:
            sc.render_raw(fr, target(ref))
        for batch in (0, 2, 8):
            sc.set_sequence_batch(batch)
            got = fresh()
            sc.render_sequence_raw(frames, target(got))
            for k in ref:
                assert np.array_equal(got[k].view(np.uint32), ref[k].view(np.uint32)), (band, batch, k)
    sc.close()


@pytest.mark.parametrize("size", [9, 30, 900])
def test_skybox_face_sizes(bunny_octree, size):
    """The environment lookup scales coordinates by the face size: sizes that are not powers of two, and the reference's
    own 900 x 900 (the oracle's lookup is pinned to llvmpipe at 9, 30, 100 and 900 by tests/golden/ref_sky_sizes.npz)."""
    from oracle import oracle
    sky = scenes.procedural_skybox(size) if size == 900 else np.random.default_rng(size).integers(0, 256, size=(6, size, size, 4), dtype=np.uint8)
    sc = api.Scene.from_octree(bunny_octree)
    sc.set_skybox(sky)
    osc = oracle.OracleScene(bunny_octree.as_dict(), sky)
    u = Uniforms.from_camera(Camera(eye=(2.0, 1.0, 2.0), at=(0.0, 3.0, 0.0), aspect=1.0)).pack()      # looking up: mostly sky
    o = osc.render(u, 128, 128)
    assert (o["kind"] == api.HIT_MISS).mean() > 0.3
    compare(o, sc.render(u, 128, 128), size)
    sc.close()
