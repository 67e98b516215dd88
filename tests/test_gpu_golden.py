"""GPU parity against the reference itself: the HIP render pass (through the C ABI) compared bit-for-bit with the
outputs that the reference's own shader produced on Mesa llvmpipe (tests/golden/ref_*.npz, made by
oracle/gen_fixtures.py).  No oracle code computes anything here; oracle.gen_fixtures is imported only for the
fixture helpers (scene arrays from the stored mesh, NDC delta decoding)."""
import glob
import os

import numpy as np
import pytest

from raytracer_amd import api

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
VIEW_FILES = sorted(f for f in glob.glob(os.path.join(GOLDEN, "ref_*.npz")) if os.path.basename(f) not in ("ref_accum.npz", "ref_accum8.npz", "ref_math.npz", "ref_sky_sizes.npz", "ref_present.npz") and not os.path.basename(f).startswith(("ref_ndc_", "ref_overlay_")))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


# pipeline + child order of the octree traversal: the shader's own, and nearest-first with the shader's result
# ("exact": built into the wave-wide traversal of the persistent and wavefront pipelines)
MODES = ["persistent", "wavefront", "megakernel", "persistent+exact", "wavefront+exact"]


def set_mode(sc, mode):
    pipe, _, order = mode.partition("+")
    sc.set_pipeline(pipe)
    sc.set_traversal_order("near_first_exact" if order == "exact" else ("auto" if pipe == "auto" else "reference"))


@pytest.fixture(scope="module", params=MODES)
def gpu_scene(request, bunny_octree):
    sc = api.Scene.from_octree(bunny_octree)
    set_mode(sc, request.param)
    yield sc
    sc.close()


def check(g, z, rows=None):
    sl = slice(None) if rows is None else slice(rows[0], rows[1])
    assert np.array_equal(g["kind"][sl], z["kind"])                  # primary hit kind
    assert np.array_equal(g["slot"][sl], z["slot"])                  # sphere index / triangle slot: bit-exact hit IDs
    assert np.array_equal(g["nrays"][sl], z["nrays"])                # path segments per pixel
    traced = z["nrays"] > 0          # with no segment at all (max_recursion_limit 0) there is no primary hit distance to compare
    assert np.array_equal(bits(g["t"][sl])[traced], bits(z["t"])[traced])
    assert np.array_equal(bits(g["depth"][sl]), bits(z["depth"]))
    rgb = g["rgba"][sl][..., :3]
    d = np.abs(rgb.astype(np.float64) - z["rgb"].astype(np.float64))
    assert (np.nan_to_num(d, nan=np.inf).max(-1) <= 1.0 / 255.0).mean() >= 0.999   # the stated colour tolerance ...
    assert np.array_equal(bits(rgb), bits(z["rgb"]))                                  # ... which is met with zero error
    assert not g["rgba"][sl][..., 3].any()


@pytest.mark.parametrize("path", VIEW_FILES, ids=[os.path.basename(p)[4:-4] for p in VIEW_FILES])
def test_hip_matches_llvmpipe_view(gpu_scene, path):
    z = np.load(path)
    w, h = (int(v) for v in z["size"])
    gpu_scene.set_skybox(z["sky"])
    check(gpu_scene.render(z["uniforms"], w, h), z)


SCENE_FILES = sorted(glob.glob(os.path.join(GOLDEN, "refscene_*.npz")))


@pytest.mark.parametrize("pipeline", MODES)
@pytest.mark.parametrize("path", SCENE_FILES, ids=[os.path.basename(p)[9:-4] for p in SCENE_FILES])
def test_hip_matches_llvmpipe_on_other_scenes(path, pipeline):
    """Scenes other than the bunny, stored with their arrays: every material on mesh triangles, the reference's quad mesh."""
    z = np.load(path)
    w, h = (int(v) for v in z["size"])
    sc = api.Scene.from_dict({k[6:]: z[k] for k in z.files if k.startswith("scene_")})
    sc.set_skybox(z["sky"])
    set_mode(sc, pipeline)
    ndc = None
    if "ndc_dx" in z.files:      # a frame size that is not a power of two: the rasteriser's interpolated NDC, stored as ulp deltas
        from oracle.gen_fixtures import apply_ulp_delta, closed_form_ndc
        cx, cy = closed_form_ndc(w, h)
        ndc = (apply_ulp_delta(cx, z["ndc_dx"]), apply_ulp_delta(cy, z["ndc_dy"]))
    try:
        check(sc.render(z["uniforms"], w, h, ndc=ndc), z)
        if pipeline.endswith("+exact"):
            # without the debug planes the traversal skips the resolution of the reference's slot (another kernel variant): same pixels
            g = sc.render(z["uniforms"], w, h, ndc=ndc, planes=False)
            assert np.array_equal(bits(g["rgba"][..., :3]), bits(z["rgb"])) and np.array_equal(bits(g["depth"]), bits(z["depth"]))
    finally:
        sc.close()


def test_hip_matches_llvmpipe_accumulation(gpu_scene):
    z = np.load(os.path.join(GOLDEN, "ref_accum.npz"))
    w, h = (int(v) for v in z["size"])
    gpu_scene.set_skybox(z["sky"])
    acc = None
    for k, t in enumerate(z["times"]):
        u = z["uniforms"].copy()
        u[20] = t
        acc = gpu_scene.render(u, w, h, accum_index=k + 1, color=acc, planes=False)["rgba"]
    assert np.array_equal(bits(acc), bits(z["rgba"]))


def test_hip_matches_llvmpipe_with_rasteriser_ndc(gpu_scene):
    from oracle.gen_fixtures import apply_ulp_delta, closed_form_ndc
    z = np.load(os.path.join(GOLDEN, "ref_ndc_640x512.npz"))
    w, h = (int(v) for v in z["size"])
    cx, cy = closed_form_ndc(w, h)
    nx, ny = apply_ulp_delta(cx, z["ndc_dx"]), apply_ulp_delta(cy, z["ndc_dy"])
    gpu_scene.set_skybox(z["sky"])
    check(gpu_scene.render(z["uniforms"], w, h, ndc=(nx, ny)), z, tuple(int(v) for v in z["rows"]))


def test_hip_sequence_matches_llvmpipe_accumulation(bunny_octree):
    """The four accumulated reference frames again, issued as ONE rt_render_sequence call into device memory: the frames
    are traced by one launch and blended inside the kernel, and must still equal what llvmpipe's blend unit produced."""
    import torch
    z = np.load(os.path.join(GOLDEN, "ref_accum.npz"))
    w, h = (int(v) for v in z["size"])
    sc = api.Scene.from_octree(bunny_octree)
    sc.set_skybox(z["sky"])
    sc.set_pipeline("megakernel")
    frames = []
    for k, t in enumerate(z["times"]):
        u = z["uniforms"].copy()
        u[20] = t
        frames.append(api.make_frame(u, w, h, accum_index=k + 1))
    for max_batch in (0, 1, 2):          # the library's choice (all four in one launch), no batching, pairs
        sc.set_sequence_batch(max_batch)
        color = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
        tg = api.rt_target()
        tg.memory, tg.sync, tg.color = api.RT_MEM_DEVICE, 1, color.data_ptr()
        sc.render_sequence_raw(frames, tg)
        assert np.array_equal(bits(color.cpu().numpy()), bits(z["rgba"])), max_batch
    sc.close()


@pytest.mark.parametrize("pipeline", ["megakernel", "wavefront", "persistent"])
def test_hip_sequence_matches_llvmpipe_eight_frame_accumulation(bunny_octree, pipeline):
    """Eight accumulated reference frames as one sequence call: the one-lane-per-pixel kernel traces them in one launch, in two
    groups of four waves whose samples go through the staging plane and blend_batch_kernel."""
    import torch
    z = np.load(os.path.join(GOLDEN, "ref_accum8.npz"))
    w, h = (int(v) for v in z["size"])
    sc = api.Scene.from_octree(bunny_octree)
    sc.set_skybox(z["sky"])
    sc.set_pipeline(pipeline)
    frames = []
    for k, t in enumerate(z["times"]):
        u = z["uniforms"].copy()
        u[20] = t
        frames.append(api.make_frame(u, w, h, accum_index=k + 1))
    color = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    tg = api.rt_target()
    tg.memory, tg.sync, tg.color = api.RT_MEM_DEVICE, 1, color.data_ptr()
    sc.render_sequence_raw(frames, tg)
    assert np.array_equal(bits(color.cpu().numpy()), bits(z["rgba"]))
    sc.close()
