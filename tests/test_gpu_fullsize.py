"""GPU tests at the headline size (3840x2160, the configuration bench.py measures): properties that do not need a full
CPU render -- the oracle on sampled row bands of the same frame, run-to-run determinism, band parts re-interleaving to
the full frame, and a sequence against frame-by-frame accumulation."""
import numpy as np
import pytest

from raytracer_amd import api, dist as rdist
from raytracer_amd.camera import Camera
from raytracer_amd.params import Uniforms

pytestmark = pytest.mark.gpu
W, H = 3840, 2160


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def scene4k(bunny_octree, test_sky):
    sc = api.Scene.from_octree(bunny_octree)
    sc.set_skybox(test_sky)
    yield sc
    sc.close()


@pytest.fixture(scope="module")
def frame4k(scene4k):
    u = Uniforms.from_camera(Camera(aspect=W / H), max_recursion_limit=2).pack()
    return u, scene4k.render(u, W, H)


def test_4k_rows_match_oracle(frame4k, bunny_octree, test_sky):
    from oracle import oracle
    u, g = frame4k
    osc = oracle.OracleScene(bunny_octree.as_dict(), test_sky)
    for rows in ((0, 8), (700, 712), (1078, 1090), (1500, 1508), (2152, 2160)):   # ground, spheres, bunny, sky
        o = osc.render(u, W, H, rows=rows)
        sl = slice(*rows)
        for k in ("kind", "slot", "nrays"):
            assert np.array_equal(g[k][sl], o[k][sl]), (rows, k)
        for k in ("t", "depth", "rgba"):
            assert np.array_equal(bits(g[k][sl]), bits(o[k][sl])), (rows, k)


def test_4k_is_deterministic_across_runs_and_pipelines(scene4k, frame4k):
    u, g = frame4k
    for pipe in ("auto", "megakernel", "wavefront", "persistent"):
        scene4k.set_pipeline(pipe)
        r = scene4k.render(u, W, H, planes=False)
        assert np.array_equal(bits(r["rgba"]), bits(g["rgba"])), pipe
        assert np.array_equal(bits(r["depth"]), bits(g["depth"])), pipe
    scene4k.set_pipeline("auto")


@pytest.mark.parametrize("parts", [2, 8])
def test_4k_band_parts_reinterleave_to_the_frame(scene4k, frame4k, parts):
    u, g = frame4k
    frame = np.full((H, W, 4), np.nan, np.float32)
    for part in range(parts):
        fr = api.make_frame(u, W, H, band=(16, parts, part), compact=True)
        rows = api.band_row_count(fr)
        assert rows == rdist.local_row_count(H, 16, parts, part)
        buf = np.zeros((rows, W, 4), np.float32)
        tg = api.rt_target()
        tg.memory, tg.sync, tg.color = api.RT_MEM_HOST, 1, buf.ctypes.data
        scene4k.render_raw(fr, tg)
        ys = rdist.local_to_frame_rows(H, 16, parts, part)
        frame[ys[ys >= 0]] = buf[ys >= 0]
    assert np.array_equal(bits(frame), bits(g["rgba"]))


def test_4k_sequence_equals_frame_by_frame(scene4k):
    n = 3
    us = [Uniforms.from_camera(Camera(aspect=W / H), time=k / 60.0, max_recursion_limit=2).pack() for k in range(n)]
    ref = np.zeros((H, W, 4), np.float32)
    for k in range(n):
        ref = scene4k.render(us[k], W, H, accum_index=k + 1, color=ref, planes=False)["rgba"]
    buf = np.zeros((H, W, 4), np.float32)
    tg = api.rt_target()
    tg.memory, tg.sync, tg.color = api.RT_MEM_HOST, 1, buf.ctypes.data
    scene4k.render_sequence_raw([api.make_frame(us[k], W, H, accum_index=k + 1) for k in range(n)], tg)
    assert np.array_equal(bits(buf), bits(ref))
