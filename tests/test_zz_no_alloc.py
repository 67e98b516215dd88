"""rt_render_noalloc (the entry point that allocates nothing: its kernel's cost map and start order live in a static
device array of the library) against rt_render, through the C ABI: every plane bit-identical, on calls that follow each
other on one GPU -- the second with a smaller frame, so it meets the first call's leftovers in the shared array --, with
no change in the device's free memory across the calls, and the documented size limit refused with RT_ELIMIT."""
import numpy as np
import pytest

from raytracer_amd import api
from raytracer_amd.camera import Camera
from raytracer_amd.params import Uniforms

pytestmark = pytest.mark.gpu

PLANES = (("color", 4, "float32"), ("depth", 1, "float32"), ("hit_kind", 1, "int32"), ("hit_slot", 1, "int32"), ("hit_t", 1, "float32"), ("n_rays", 1, "int32"))


def _targets(torch, w, rows, start=None):
    t = {}
    for name, ch, dt in PLANES:
        shape = (rows, w, ch) if ch > 1 else (rows, w)
        t[name] = torch.zeros(shape, dtype=getattr(torch, dt), device="cuda")
    if start is not None:
        t["color"].copy_(start)
    tg = api.rt_target()
    tg.memory, tg.sync = api.RT_MEM_DEVICE, 1
    for name, _, _ in PLANES:
        setattr(tg, name, t[name].data_ptr())
    return t, tg


def _same(torch, a, b, what):
    for name, _, _ in PLANES:
        assert torch.equal(a[name].view(torch.int32), b[name].view(torch.int32)), (what, name)


def test_noalloc_matches_rt_render_on_consecutive_calls(bunny_octree, test_sky):
    import torch
    sc = api.Scene.from_octree(bunny_octree, device=0)
    sc.set_skybox(test_sky)
    sc.set_pipeline("megakernel")
    stream = torch.cuda.current_stream().cuda_stream
    rng = torch.Generator(device="cuda").manual_seed(5)
    # (frame, band, compact): a large frame, then smaller ones that find its costs and order in the array, a band of a frame,
    # the first geometry again (its order is gone by then), and the same geometry twice in a row (the second call uses
    # the order the first one left)
    cases = [(1280, 720, (16, 1, 0), False), (320, 200, (16, 1, 0), False), (333, 77, (16, 1, 0), False), (640, 360, (16, 3, 1), True),
             (1280, 720, (16, 1, 0), False), (1280, 720, (16, 1, 0), False), (1280, 720, (16, 1, 0), False), (64, 64, (16, 1, 0), False)]
    torch.cuda.synchronize()
    free_before = None
    for i, (w, h, band, compact) in enumerate(cases):
        u = Uniforms.from_camera(Camera(aspect=w / h), time=i / 60.0, max_recursion_limit=4).pack()
        fr = api.make_frame(u, w, h, accum_index=i + 1, blend=True, band=band, stream=stream, compact=compact)
        rows = api.band_row_count(fr) if compact else h
        start = torch.rand((rows, w, 4), generator=rng, device="cuda")
        ref, tg_ref = _targets(torch, w, rows, start)
        sc.render_raw(fr, tg_ref)
        got, tg = _targets(torch, w, rows, start)
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        sc.render_noalloc_raw(fr, tg)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free0, "device memory changed during rt_render_noalloc"
        if free_before is None:
            free_before = free0
        _same(torch, ref, got, (i, w, h, band))
        assert int((got["hit_kind"] == api.HIT_TRIANGLE).sum()) > 0
    sc.close()


def test_noalloc_with_counters_async_and_two_scenes_sharing_the_array(bunny_octree, test_sky):
    """sync = 0 calls in a row on one stream (every call reads the order its predecessor wrote), a second scene on the same
    GPU in between (same array), and the work counters."""
    import torch
    from raytracer_amd import scenes
    a = api.Scene.from_octree(bunny_octree, device=0)
    b = api.Scene.from_octree(scenes.small_test_scene(), device=0)
    for sc in (a, b):
        sc.set_skybox(test_sky)
        sc.set_pipeline("megakernel")
    stream = torch.cuda.current_stream().cuda_stream
    w, h = 512, 288
    us = [Uniforms.from_camera(Camera(aspect=w / h), time=k / 60.0, max_recursion_limit=3).pack() for k in range(10)]
    want, got = {}, {}
    for sc, name in ((a, "a"), (b, "b")):
        col = torch.zeros((h, w, 4), device="cuda")
        tg = api.rt_target()
        tg.memory, tg.sync, tg.color = api.RT_MEM_DEVICE, 1, col.data_ptr()
        for k, u in enumerate(us):
            sc.render_raw(api.make_frame(u, w, h, accum_index=k + 1, stream=stream), tg)
        want[name] = col
        got[name] = torch.zeros((h, w, 4), device="cuda")
    for k, u in enumerate(us):
        for sc, name in ((a, "a"), (b, "b")):
            tg = api.rt_target()
            tg.memory, tg.sync, tg.color = api.RT_MEM_DEVICE, 0, got[name].data_ptr()
            sc.render_noalloc_raw(api.make_frame(u, w, h, accum_index=k + 1, stream=stream), tg)
    torch.cuda.synchronize()
    for name in ("a", "b"):
        assert torch.equal(want[name].view(torch.int32), got[name].view(torch.int32)), name
    a.set_counting(True)
    col = torch.zeros((h, w, 4), device="cuda")
    tg = api.rt_target()
    tg.memory, tg.sync, tg.color = api.RT_MEM_DEVICE, 1, col.data_ptr()
    a.render_raw(api.make_frame(us[0], w, h, stream=stream), tg)
    st_ref = a.stats()
    a.render_noalloc_raw(api.make_frame(us[0], w, h, stream=stream), tg)
    assert a.stats() == st_ref and st_ref["rays"] > w * h
    a.close(); b.close()


def test_noalloc_rejects_what_it_cannot_do(bunny_octree):
    import torch
    sc = api.Scene.from_octree(bunny_octree, device=0)
    u = Uniforms.from_camera(Camera(aspect=1.0)).pack()
    tg = api.rt_target()
    # more than 2^17 tiles of 16x16 pixels in one call: 16384 x 2064 pixels = 1024 x 129 tiles
    w, h = 16384, 2064
    col = torch.zeros((16, 16, 4), device="cuda")      # never written: the call must fail before any launch
    tg.memory, tg.sync, tg.color = api.RT_MEM_DEVICE, 1, col.data_ptr()
    with pytest.raises(api.RtError) as e:
        sc.render_noalloc_raw(api.make_frame(u, w, h), tg)
    assert e.value.code == api.RT_ELIMIT and "tiles" in str(e.value)
    host = np.zeros((32, 32, 4), np.float32)
    tg.memory, tg.color = api.RT_MEM_HOST, host.ctypes.data
    with pytest.raises(api.RtError) as e:
        sc.render_noalloc_raw(api.make_frame(u, 32, 32), tg)
    assert e.value.code == api.RT_EINVAL
    # the same oversized frame in two bands fits
    fr = api.make_frame(u, 16384, 2064, band=(16, 2, 0), compact=True)
    rows = api.band_row_count(fr)
    band = torch.zeros((rows, 16384, 4), device="cuda")
    tg.memory, tg.color = api.RT_MEM_DEVICE, band.data_ptr()
    sc.render_noalloc_raw(fr, tg)
    ref = torch.zeros((rows, 16384, 4), device="cuda")
    tg.color = ref.data_ptr()
    sc.render_raw(fr, tg)
    assert torch.equal(ref.view(torch.int32), band.view(torch.int32))
    sc.close()


def test_noalloc_captured_call_replays_after_another_geometry_used_the_array(bunny_octree, test_sky):
    """A call captured into a HIP graph is replayed after calls with OTHER frame geometries have rewritten the shared array's cost
    map and start order: a captured call uses neither (its tiles start in index order), so every replay gives rt_render's frame."""
    import torch
    sc = api.Scene.from_octree(bunny_octree, device=0)
    sc.set_skybox(test_sky)
    sc.set_pipeline("megakernel")
    w, h = 640, 368
    u = Uniforms.from_camera(Camera(aspect=w / h), max_recursion_limit=3).pack()
    # (first calls outside any capture: they load the kernels' code and leave an order for this geometry in the array)
    warm, tg_warm = _targets(torch, w, h)
    for _ in range(2):
        sc.render_noalloc_raw(api.make_frame(u, w, h, stream=torch.cuda.current_stream().cuda_stream), tg_warm)
    torch.cuda.synchronize()
    got, tg = _targets(torch, w, h)
    tg.sync = 0
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        sc.render_noalloc_raw(api.make_frame(u, w, h, blend=False, stream=torch.cuda.current_stream().cuda_stream), tg)
    # other geometries in between: a smaller frame twice (the second call sorts a new order into the array), a band of a larger one
    for (ow, oh, band, compact) in ((320, 192, (16, 1, 0), False), (320, 192, (16, 1, 0), False), (1280, 720, (16, 2, 1), True)):
        ou = Uniforms.from_camera(Camera(aspect=ow / oh), max_recursion_limit=2).pack()
        fr = api.make_frame(ou, ow, oh, band=band, compact=compact, stream=torch.cuda.current_stream().cuda_stream)
        other, tg_other = _targets(torch, ow, api.band_row_count(fr) if compact else oh)
        sc.render_noalloc_raw(fr, tg_other)
    torch.cuda.synchronize()
    ref, tg_ref = _targets(torch, w, h)
    sc.render_raw(api.make_frame(u, w, h, blend=False, stream=torch.cuda.current_stream().cuda_stream), tg_ref)
    for _ in range(3):
        for name, _, _ in PLANES:
            got[name].zero_()
        graph.replay()
        torch.cuda.synchronize()
        _same(torch, ref, got, "replay")
    assert int((got["hit_kind"] == api.HIT_TRIANGLE).sum()) > 0
    # and an uncaptured call afterwards still works with the array (its bookkeeping was not touched by the capture)
    again, tg_again = _targets(torch, w, h)
    sc.render_noalloc_raw(api.make_frame(u, w, h, blend=False, stream=torch.cuda.current_stream().cuda_stream), tg_again)
    _same(torch, ref, again, "after the replays")
    sc.close()
