"""Whole frames at the benchmark's sizes against the reference shader on llvmpipe: tests/golden/ref_fullframe_digests.json (made by
`python -m oracle.gen_fixtures fullframe`) holds the SHA-256 of every plane of the reference's 1280x720, 3840x2160 and 7680x4320 frames -- the
benchmark's camera on the bunny scene and on the two seeded synthetic scenes -- since the planes themselves are too large to keep.  The
HIP pass renders the same frames with the rasteriser's interpolated NDC (ref_ndc_<w>x<h>.npz) and must reproduce every digest: primary
hit kind, triangle slot, distance, segment count, depth and colour of all 0.9 / 8.3 / 33.2 million pixels, in the library's default mode and
with pipeline and child order pinned."""
import hashlib
import json
import os

import numpy as np
import pytest

from raytracer_amd import api, scenes

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ENTRIES = json.load(open(os.path.join(GOLDEN, "ref_fullframe_digests.json")))
BUILDERS = {"bunny": scenes.default_octree, "synthetic_100k": scenes.synthetic_100k, "synthetic_1m": scenes.synthetic_1m}


def plane_digests(g):
    out = {}
    for k, src, dt in (("rgb", g["rgba"][..., :3], np.float32), ("depth", g["depth"], np.float32), ("t", g["t"], np.float32), ("slot", g["slot"], np.uint32),
                       ("kind", g["kind"], np.uint8), ("nrays", g["nrays"], np.uint8)):
        out[k] = hashlib.sha256(np.ascontiguousarray(src, dtype=dt).tobytes()).hexdigest()
    return out


@pytest.mark.parametrize("name", sorted(ENTRIES))
def test_whole_frame_matches_the_reference_digests(name):
    from oracle.gen_fixtures import apply_ulp_delta, closed_form_ndc, fixture_sky, scene_digests
    e = ENTRIES[name]
    w, h = e["width"], e["height"]
    oc = BUILDERS[e["scene"]](device=0)
    d = oc.as_dict()
    assert scene_digests(d) == e["scene_sha256"]          # the scene the reference rendered
    z = np.load(os.path.join(GOLDEN, f"ref_ndc_{w}x{h}.npz"))
    cx, cy = closed_form_ndc(w, h)
    ndc = (apply_ulp_delta(cx, z["ndc_dx"]), apply_ulp_delta(cy, z["ndc_dy"]))
    u = np.array(e["uniforms_bits"], np.uint32).view(np.float32)
    sc = api.Scene.from_dict(d, device=0)
    sc.set_skybox(fixture_sky())
    modes = [("auto", "auto")] + ([("wavefront", "near_first_exact"), ("persistent", "near_first_exact"), ("wavefront", "reference")] if e["scene"] != "bunny" else [("wavefront", "near_first_exact"), ("persistent", "reference")])
    try:
        for pipe, order in modes:
            sc.set_pipeline(pipe); sc.set_traversal_order(order)
            g = sc.render(u, w, h, ndc=ndc)
            assert int(g["nrays"].astype(np.int64).sum()) == e["rays"], (pipe, order)
            assert np.bincount(g["kind"].ravel(), minlength=6).tolist() == e["kinds"], (pipe, order)
            assert plane_digests(g) == e["sha256"], (pipe, order)
            assert not g["rgba"][..., 3].any()
    finally:
        sc.close()
