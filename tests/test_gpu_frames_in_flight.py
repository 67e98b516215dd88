"""rt_render_sequence with several frames in flight (rt_scene_set_frames_in_flight): frames traced concurrently into a staging
plane and blended in frame order must leave exactly what frame-after-frame rendering leaves — colour accumulation
(App.cpp:49,102-105: the blend state of the reference's draw) and, in the other planes, the last frame's values."""
import numpy as np
import pytest

from raytracer_amd import api, dist as rdist
from raytracer_amd.camera import Camera
from raytracer_amd.params import Uniforms

pytestmark = pytest.mark.gpu

PLANES = ("color", "depth", "kind", "slot", "t", "nrays")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def run_sequence(sc, frames, rows, w, pattern):
    got = dict(color=np.full((rows, w, 4), pattern, np.float32), depth=np.full((rows, w), 0.75, np.float32), kind=np.full((rows, w), 77, np.uint32),
               slot=np.full((rows, w), 123456, np.uint32), t=np.full((rows, w), -3.0, np.float32), nrays=np.full((rows, w), 55, np.uint32))
    tg = api.rt_target()
    tg.memory, tg.sync = api.RT_MEM_HOST, 1
    tg.color, tg.depth, tg.hit_kind, tg.hit_slot, tg.hit_t, tg.n_rays = (got[k].ctypes.data for k in PLANES)
    sc.render_sequence_raw(frames, tg)
    return got


@pytest.mark.parametrize("pipe", ["wavefront", "persistent"])
def test_frames_in_flight_equal_serial_sequence(bunny_octree, test_sky, pipe):
    sc = api.Scene.from_octree(bunny_octree)
    sc.set_skybox(test_sky)
    sc.set_pipeline(pipe)
    w, h, n = 1400, 800, 9
    blends = [True, True, False, True, True, True, True, False, True]
    us = [Uniforms.from_camera(Camera(aspect=w / h), time=0.37 * k, max_recursion_limit=1 + k % 4).pack() for k in range(n)]
    for band, compact in (((16, 1, 0), False), ((16, 3, 1), True), ((7, 2, 1), True)):
        frames = [api.make_frame(us[k], w, h, accum_index=5 + k, blend=blends[k], band=band, compact=compact) for k in range(n)]
        rows = api.band_row_count(frames[0]) if compact else h
        sc.set_subframes(1); sc.set_frames_in_flight(1)
        want = run_sequence(sc, frames, rows, w, 0.125)
        for sub, flight in ((0, 0), (1, 2), (1, 3), (1, 8), (2, 2), (2, 4), (3, 0)):
            sc.set_subframes(sub); sc.set_frames_in_flight(flight)
            got = run_sequence(sc, frames, rows, w, 0.125)
            for k in PLANES:
                assert np.array_equal(bits(got[k]), bits(want[k])), (pipe, band, sub, flight, k)
        # a sequence shorter than the frames in flight, and one frame more than a multiple of them
        for m in (2, 4):
            sc.set_subframes(1); sc.set_frames_in_flight(1)
            want = run_sequence(sc, frames[:m], rows, w, 0.5)
            sc.set_subframes(0); sc.set_frames_in_flight(3)
            got = run_sequence(sc, frames[:m], rows, w, 0.5)
            for k in PLANES:
                assert np.array_equal(bits(got[k]), bits(want[k])), (pipe, band, m, k)
    sc.close()


def test_frames_in_flight_device_target_back_to_back(bunny_octree, test_sky):
    """Unsynchronised calls on one stream, each continuing the accumulation of the one before, with a frame geometry that
    changes in between (the staging plane and the workspaces are outgrown while earlier work may still read them)."""
    import torch
    sc = api.Scene.from_octree(bunny_octree)
    sc.set_skybox(test_sky)
    sc.set_pipeline("wavefront")
    stream = torch.cuda.current_stream().cuda_stream
    results = {}
    for flight in (1, 3):
        sc.set_subframes(1); sc.set_frames_in_flight(flight)
        outs = []
        for w, h in ((320, 200), (640, 360), (1280, 720)):
            color = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
            tg = api.rt_target()
            tg.memory, tg.sync, tg.color = api.RT_MEM_DEVICE, 0, color.data_ptr()
            for call in range(3):
                frames = [api.make_frame(Uniforms.from_camera(Camera(aspect=w / h), time=0.1 * (5 * call + k), max_recursion_limit=3).pack(), w, h,
                                         accum_index=5 * call + k + 1, stream=stream) for k in range(5)]
                sc.render_sequence_raw(frames, tg)
            outs.append(color)
        torch.cuda.synchronize()
        results[flight] = [o.cpu().numpy() for o in outs]
    for a, b in zip(results[1], results[3]):
        assert np.array_equal(bits(a), bits(b))
    sc.close()


def test_stacked_frames_equal_serial_sequence(bunny_octree, test_sky):
    """Frames that differ in `time` only share one chain of wavefront launches (rt_scene_set_frames_per_launch) when colour is
    the only plane asked for; runs of such frames are broken by a change of the recursion limit, of the camera, and by the
    sequence's end."""
    sc = api.Scene.from_octree(bunny_octree)
    sc.set_skybox(test_sky)
    sc.set_pipeline("wavefront")
    w, h, n = 1000, 600, 13
    cams = [Camera(aspect=w / h), Camera(eye=(2.5, 1.5, 1.0), at=(0.0, 0.5, 0.0), aspect=w / h)]
    us = [Uniforms.from_camera(cams[1 if k in (7, 8) else 0], time=0.21 * k, max_recursion_limit=2 if k == 4 else 4).pack() for k in range(n)]
    blends = [k != 5 for k in range(n)]
    for band, compact in (((16, 1, 0), False), ((16, 4, 2), True), ((5, 2, 0), True)):
        frames = [api.make_frame(us[k], w, h, accum_index=3 + k, blend=blends[k], band=band, compact=compact) for k in range(n)]
        rows = api.band_row_count(frames[0]) if compact else h

        def colour(sub, flight, stack, m=n):
            sc.set_subframes(sub); sc.set_frames_in_flight(flight); sc.set_frames_per_launch(stack)
            buf = np.full((rows, w, 4), 0.25, np.float32)
            tg = api.rt_target()
            tg.memory, tg.sync, tg.color = api.RT_MEM_HOST, 1, buf.ctypes.data
            sc.render_sequence_raw(frames[:m], tg)
            return buf

        want = colour(1, 1, 1)
        for sub, flight, stack in ((0, 0, 0), (1, 3, 2), (1, 2, 8), (1, 3, 3), (2, 2, 4), (1, 8, 1)):
            assert np.array_equal(bits(colour(sub, flight, stack)), bits(want)), (band, sub, flight, stack)
        for m in (2, 3, 5):
            assert np.array_equal(bits(colour(0, 0, 0, m)), bits(colour(1, 1, 1, m))), (band, m)
    sc.close()


def test_frames_in_flight_argument_range(bunny_octree):
    sc = api.Scene.from_octree(bunny_octree)
    for bad in (-1, 9):
        with pytest.raises(api.RtError):
            sc.set_frames_in_flight(bad)
    sc.set_frames_in_flight(0)
    for bad in (-1, 9):
        with pytest.raises(api.RtError):
            sc.set_frames_per_launch(bad)
    sc.set_frames_per_launch(0)
    sc.close()
