"""CPU tests: the oracle (oracle/rt_oracle.c) against the golden outputs of the reference shader on llvmpipe.

The fixtures under tests/golden/ref_*.npz were produced by oracle/gen_fixtures.py from the reference's own shader
text run on Mesa llvmpipe; these tests pin the oracle to them bit-for-bit."""
import glob
import os

import numpy as np
import pytest

from oracle import oracle
from oracle.gen_fixtures import apply_ulp_delta, closed_form_ndc, bunny_scene

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
VIEW_FILES = sorted(f for f in glob.glob(os.path.join(GOLDEN, "ref_*.npz")) if os.path.basename(f) not in ("ref_accum.npz", "ref_accum8.npz", "ref_math.npz", "ref_sky_sizes.npz", "ref_present.npz") and not os.path.basename(f).startswith(("ref_ndc_", "ref_overlay_")))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def scene():
    return bunny_scene()


def check_record(o, z, rows=None):
    sl = slice(None) if rows is None else slice(rows[0], rows[1])
    assert np.array_equal(o["kind"][sl], z["kind"])
    assert np.array_equal(o["slot"][sl], z["slot"])
    assert np.array_equal(o["nrays"][sl], z["nrays"])
    traced = z["nrays"] > 0          # with no segment at all (max_recursion_limit 0) there is no primary hit distance to compare
    assert np.array_equal(bits(o["t"][sl])[traced], bits(z["t"])[traced])
    assert np.array_equal(bits(o["seed"][sl]), bits(z["seed"]))
    assert np.array_equal(bits(o["depth"][sl]), bits(z["depth"]))
    assert np.array_equal(bits(o["rgba"][sl][..., :3]), bits(z["rgb"]))
    assert float(z["alpha_max"]) == 0.0 and not o["rgba"][sl][..., 3].any()


@pytest.mark.parametrize("path", VIEW_FILES, ids=[os.path.basename(p)[4:-4] for p in VIEW_FILES])
def test_oracle_matches_reference_view(scene, path):
    z = np.load(path)
    w, h = (int(v) for v in z["size"])
    o = oracle.OracleScene(scene, z["sky"]).render(z["uniforms"], w, h)
    check_record(o, z)


SCENE_FILES = sorted(glob.glob(os.path.join(GOLDEN, "refscene_*.npz")))


def scene_of(z):
    return {k[6:]: z[k] for k in z.files if k.startswith("scene_")}


@pytest.mark.parametrize("path", SCENE_FILES, ids=[os.path.basename(p)[9:-4] for p in SCENE_FILES])
def test_oracle_matches_reference_on_other_scenes(path):
    """Reference records of scenes other than the bunny (all seven materials on mesh triangles; the reference's quad mesh),
    stored with their scene arrays."""
    z = np.load(path)
    w, h = (int(v) for v in z["size"])
    ndc = None
    if "ndc_dx" in z.files:      # a frame size that is not a power of two: the rasteriser's interpolated NDC, stored as ulp deltas
        cx, cy = closed_form_ndc(w, h)
        ndc = (apply_ulp_delta(cx, z["ndc_dx"]), apply_ulp_delta(cy, z["ndc_dy"]))
    o = oracle.OracleScene(scene_of(z), z["sky"]).render(z["uniforms"], w, h, ndc=ndc)
    check_record(o, z)


def test_scene_fixture_coverage():
    assert len(SCENE_FILES) >= 5
    mats = set()
    for p in SCENE_FILES:
        z = np.load(p)
        tri = z["kind"] == 2
        if os.path.basename(p) == "refscene_empty.npz":
            assert tri.sum() == 0
            continue
        assert tri.sum() > 1000
        mats |= set(np.unique(z["scene_tri_refs"].reshape(-1, 4)[z["slot"][tri], 3]).tolist())
    assert mats == {0, 1, 2, 3, 4, 5, 6}


def test_fixture_coverage():
    """The views together exercise every hit kind and every path length up to the recursion limit."""
    kinds, nrays = set(), set()
    for p in VIEW_FILES:
        z = np.load(p)
        kinds |= set(np.unique(z["kind"]).tolist())
        nrays |= set(np.unique(z["nrays"]).tolist())
    assert kinds == {0, 1, 2, 3, 4, 5}
    assert {1, 2, 3, 4, 5, 8} <= nrays


def test_oracle_progressive_accumulation(scene):
    z = np.load(os.path.join(GOLDEN, "ref_accum.npz"))
    w, h = (int(v) for v in z["size"])
    osc = oracle.OracleScene(scene, z["sky"])
    acc = None
    for k, t in enumerate(z["times"]):
        u = z["uniforms"].copy()
        u[20] = t
        acc = osc.render(u, w, h, accum_index=k + 1, rgba=acc)["rgba"]
    assert np.array_equal(bits(acc), bits(z["rgba"]))


def test_oracle_with_interpolated_ndc_table(scene):
    """At the reference's own window size (640x512) the rasteriser's varying differs from the closed form on most
    pixels; fed the captured table, the oracle reproduces the reference on the stored band of rows."""
    z = np.load(os.path.join(GOLDEN, "ref_ndc_640x512.npz"))
    w, h = (int(v) for v in z["size"])
    cx, cy = closed_form_ndc(w, h)
    nx, ny = apply_ulp_delta(cx, z["ndc_dx"]), apply_ulp_delta(cy, z["ndc_dy"])
    assert (z["ndc_dx"] != 0).mean() > 0.5 and not z["ndc_dy"].any()
    rows = tuple(int(v) for v in z["rows"])
    o = oracle.OracleScene(scene, z["sky"]).render(z["uniforms"], w, h, rows=rows, ndc=(nx, ny))
    check_record(o, z, rows)
    # and the closed form does NOT reproduce it (the seed hashes the NDC bits), which is why the table exists
    o2 = oracle.OracleScene(scene, z["sky"]).render(z["uniforms"], w, h, rows=rows)
    assert not np.array_equal(bits(o2["seed"][rows[0]:rows[1]]), bits(z["seed"]))


@pytest.mark.parametrize("size", [(1280, 720), (3840, 2160), (7680, 4320)])
def test_oracle_with_interpolated_ndc_at_the_benchmark_frame_sizes(scene, size):
    """The same at 1280x720, 3840x2160 and 7680x4320, where x AND y differ from the closed form on most pixels."""
    w, h = size
    z = np.load(os.path.join(GOLDEN, f"ref_ndc_{w}x{h}.npz"))
    cx, cy = closed_form_ndc(w, h)
    nx, ny = apply_ulp_delta(cx, z["ndc_dx"]), apply_ulp_delta(cy, z["ndc_dy"])
    assert (z["ndc_dx"] != 0).mean() > 0.5 and (z["ndc_dy"] != 0).mean() > 0.5
    rows = tuple(int(v) for v in z["rows"])
    o = oracle.OracleScene(scene, z["sky"]).render(z["uniforms"], w, h, rows=rows, ndc=(nx, ny))
    check_record(o, z, rows)


SYNTH_FILES = sorted(glob.glob(os.path.join(GOLDEN, "refsynth_*.npz")))


@pytest.fixture(scope="module")
def synthetic_scenes():
    """The two seeded synthetic scenes, built once by the host builder (the fixtures hold digests, not arrays)."""
    from raytracer_amd import scenes
    return {"100k": scenes.synthetic_100k().as_dict(), "1m": scenes.synthetic_1m().as_dict()}


@pytest.mark.parametrize("path", SYNTH_FILES, ids=[os.path.basename(p)[9:-4] for p in SYNTH_FILES])
def test_oracle_matches_reference_on_the_synthetic_scenes(synthetic_scenes, path):
    from oracle.gen_fixtures import scene_digests
    z = np.load(path)
    scene = synthetic_scenes[os.path.basename(path).split("_")[1]]
    for k, v in scene_digests(scene).items():
        assert str(z["sha256_" + k]) == v, k
    w, h = (int(v) for v in z["size"])
    o = oracle.OracleScene(scene, z["sky"]).render(z["uniforms"], w, h)
    check_record(o, z)


def test_synthetic_fixture_coverage():
    assert len(SYNTH_FILES) == 6
    for p in SYNTH_FILES:
        z = np.load(p)
        assert (z["kind"] == 2).sum() > 900 and "scene_vertices" not in z.files
    z = np.load(os.path.join(GOLDEN, "refsynth_1m_bench.npz"))
    assert int(z["n_ref"]) > 8_000_000 and int(z["nrays"].max()) == 5 and (z["nrays"] == 5).mean() > 0.5


def test_oracle_math_matches_llvmpipe():
    z = np.load(os.path.join(GOLDEN, "ref_math.npz"))
    assert np.array_equal(bits(oracle.math("sin", z["x"])), bits(z["sin"]))
    assert np.array_equal(bits(oracle.math("cos", z["x"])), bits(z["cos"]))
    third = np.full_like(z["p"], np.float32(1.0) / np.float32(3.0))
    assert np.array_equal(bits(oracle.math("pow", z["p"], third)), bits(z["cbrt"]))
    assert np.array_equal(bits(oracle.math("pow", z["p"], np.full_like(z["p"], 5.0))), bits(z["pow5"]))


def test_oracle_cubemap_matches_llvmpipe():
    z = np.load(os.path.join(GOLDEN, "ref_math.npz"))
    empty = dict(vertices=np.zeros((1, 4)), normals=np.zeros((1, 4)), tri_refs=np.zeros((1, 4)), nodes=np.zeros(0), bounds_min=[0, 0, 0], bounds_max=[0, 0, 0])
    o = oracle.OracleScene(empty, z["sky"]).sample_sky(z["dirs"])
    assert np.array_equal(bits(o), bits(z["sky_rgb"]))


@pytest.mark.parametrize("size", [9, 30, 100, 900])
def test_oracle_cubemap_at_other_face_sizes(size):
    z = np.load(os.path.join(GOLDEN, "ref_sky_sizes.npz"))
    empty = dict(vertices=np.zeros((1, 4)), normals=np.zeros((1, 4)), tri_refs=np.zeros((1, 4)), nodes=np.zeros(0), bounds_min=[0, 0, 0], bounds_max=[0, 0, 0])
    if size == 900:      # the reference's own face size: faces regenerated from their seed, pinned by digest
        import hashlib
        from raytracer_amd import scenes
        sky = scenes.procedural_skybox(900)
        assert hashlib.sha256(np.ascontiguousarray(sky).tobytes()).digest() == z["sky_900_sha256"].tobytes()
    else:
        sky = z[f"sky_{size}"]
    o = oracle.OracleScene(empty, sky).sample_sky(z["dirs"])
    assert np.array_equal(bits(o), bits(z[f"rgb_{size}"]))


def test_oracle_eight_frame_accumulation(scene):
    z = np.load(os.path.join(GOLDEN, "ref_accum8.npz"))
    w, h = (int(v) for v in z["size"])
    osc = oracle.OracleScene(scene, z["sky"])
    acc = None
    for k, t in enumerate(z["times"]):
        u = z["uniforms"].copy()
        u[20] = t
        acc = osc.render(u, w, h, accum_index=k + 1, rgba=acc, taps=False)["rgba"]
    assert np.array_equal(bits(acc), bits(z["rgba"]))


ATTACHMENT_FILES = sorted(f for f in glob.glob(os.path.join(GOLDEN, "ref*.npz")) if "attachment_depth" in np.load(f).files)


def attachment_values(z):
    """24-bit values behind the GL_FLOAT read-back of the reference's depth attachment (value / (2^24 - 1) as a float)."""
    a = z["attachment_depth"]
    v = np.rint(a.astype(np.float64) * 16777215.0).astype(np.uint32)
    assert np.array_equal(bits((v.astype(np.float64) / 16777215.0).astype(np.float32)), bits(a))   # the read-back is exactly such a value
    return v


@pytest.mark.parametrize("path", ATTACHMENT_FILES, ids=[os.path.basename(p)[:-4] for p in ATTACHMENT_FILES])
def test_depth_attachment_values_match_llvmpipe(path):
    """What the 24-bit depth attachment holds after the reference's draw, against the float the shader wrote."""
    z = np.load(path)
    assert np.array_equal(oracle.depth_attachment(z["depth"]), attachment_values(z))


def test_depth_attachment_fixture_coverage():
    assert len(ATTACHMENT_FILES) >= 20
    seen = np.unique(np.concatenate([attachment_values(np.load(p)).ravel() for p in ATTACHMENT_FILES]))
    assert len(seen) > 100000 and seen.max() == 16777215 and seen.min() == 0 and (seen < 8388608).sum() > 2000      # depths below 0.5, where the float product rounds to halves
    assert np.array_equal(oracle.depth_attachment(np.array([-1.0, 0.0, 0.5, 1.0, 2.0, np.nan], np.float32)), [0, 0, 8388608, 16777215, 16777215, 0])


@pytest.mark.parametrize("name", ["bunny_720p_limit5", "bunny_4k_limit2", "bunny_8k_limit5"])
def test_oracle_matches_the_reference_whole_frame_digests(scene, name):
    """The reference's WHOLE 1280x720, 3840x2160 and 7680x4320 frames of the bunny scene, kept as SHA-256 digests of their planes
    (ref_fullframe_digests.json): the restatement reproduces every one of them with the rasteriser's NDC table."""
    import hashlib
    import json
    from oracle.gen_fixtures import fixture_sky
    e = json.load(open(os.path.join(GOLDEN, "ref_fullframe_digests.json")))[name]
    w, h = e["width"], e["height"]
    z = np.load(os.path.join(GOLDEN, f"ref_ndc_{w}x{h}.npz"))
    cx, cy = closed_form_ndc(w, h)
    ndc = (apply_ulp_delta(cx, z["ndc_dx"]), apply_ulp_delta(cy, z["ndc_dy"]))
    u = np.array(e["uniforms_bits"], np.uint32).view(np.float32)
    o = oracle.OracleScene(scene, fixture_sky()).render(u, w, h, ndc=ndc)
    for k, src, dt in (("rgb", o["rgba"][..., :3], np.float32), ("depth", o["depth"], np.float32), ("t", o["t"], np.float32), ("slot", o["slot"], np.uint32),
                       ("kind", o["kind"], np.uint8), ("nrays", o["nrays"], np.uint8)):
        assert hashlib.sha256(np.ascontiguousarray(src, dtype=dt).tobytes()).hexdigest() == e["sha256"][k], k
