"""Test infrastructure, run by hand where /root/reference exists:  python -m oracle.fuzz_vs_reference [--views N] [--seed S] [--scene bunny|small|planes|deep|s100k|s1m]

Random views (the generator of tests/fuzz_parity.py, frame sizes forced to powers of two so that the closed-form NDC
equals the rasteriser's) rendered by the reference's own shader on llvmpipe (oracle/refrun.py) and by the CPU restatement
(oracle/rt_oracle.c); colour, the shader's depth, hit kind, slot, distance and segment count are compared bit for bit."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=40)
    ap.add_argument("--seed", type=int, default=101)
    ap.add_argument("--any-size", action="store_true", help="frame sizes 1..200 in both directions; the rasteriser's interpolated NDC of each size is captured from llvmpipe and handed to the restatement as a table")
    ap.add_argument("--scene", default="bunny", choices=["bunny", "small", "planes", "deep", "s100k", "s1m"],
                    help="small: a cluster of meshes carrying all seven materials on triangles; planes: sheets with vertices, edges and corners on the octree's cell planes, "
                         "half of the eyes on cell planes too; deep: a 26-level tree; s100k / s1m: the seeded 100 000-primitive / 1.28 M-triangle scenes")
    ap.add_argument("--eye-in-plane", action="store_true", help="two thirds of the views with the eye IN the plane of a random scene triangle, looking along that plane (tests/fuzz_parity.py in_plane_view)")
    a = ap.parse_args()
    from fuzz_parity import random_view, in_plane_view
    from oracle import oracle, refrun
    from oracle.gen_fixtures import bunny_scene
    refrun.build()
    assert refrun.available(), "needs /root/reference and the llvmpipe runner"
    rng = np.random.default_rng(a.seed)
    if a.scene == "bunny":
        scene = bunny_scene()
    else:
        from raytracer_amd import scenes          # (host-side octree builder of the product; its arrays equal the octree oracle's)
        scene = {"small": scenes.small_test_scene, "planes": scenes.plane_aligned_scene, "deep": scenes.deep_test_scene,
                 "s100k": scenes.synthetic_100k, "s1m": scenes.synthetic_1m}[a.scene]().as_dict()
    sky = rng.integers(0, 256, size=(6, 48, 48, 4), dtype=np.uint8)
    osc = oracle.OracleScene(scene, sky)
    bad = 0
    ndc_cache = {}
    for v in range(a.views):
        u, _, _ = random_view(rng, lattice=a.scene == "planes")
        w, h = [(128, 128), (128, 64), (64, 128)][int(rng.integers(0, 3))]
        in_plane_ndc = None
        if a.eye_in_plane and rng.uniform() < 0.67:
            pv = in_plane_view(rng, np.asarray(scene["vertices"]).reshape(-1, 4), np.asarray(scene["tri_refs"]).reshape(-1, 4))
            if pv is not None:      # (its own, mostly odd, frame size: the rasteriser's NDC of that size is captured like with --any-size)
                from oracle.gen_fixtures import capture_ndc
                u, w, h = pv
                if (w, h) not in ndc_cache:
                    ndc_cache[(w, h)] = capture_ndc(w, h)
                in_plane_ndc = ndc_cache[(w, h)]
        # the aspect ratio baked into the matrix no longer matches the frame: any matrix is a valid input
        ndc = in_plane_ndc
        if a.any_size and in_plane_ndc is None:
            from oracle.gen_fixtures import capture_ndc
            w, h = int(rng.integers(1, 201)), int(rng.integers(1, 201))
            ndc = capture_ndc(w, h)
        col = refrun.run_reference(scene, u, w, h, sky=sky)
        hit = refrun.run_reference(scene, u, w, h, sky=sky, tap="hit")
        o = osc.render(u, w, h, ndc=ndc)
        traced = hit["nrays"] > 0
        diffs = {"rgb": int((o["rgba"][..., :3].view(np.uint32) != col["rgba"][..., :3].view(np.uint32)).sum()),
                 "kind": int((o["kind"] != hit["kind"]).sum()), "slot": int((o["slot"] != hit["slot"]).sum()), "nrays": int((o["nrays"] != hit["nrays"]).sum()),
                 "t": int((o["t"].view(np.uint32)[traced] != hit["t"].view(np.uint32)[traced]).sum()),
                 "depth24": int((oracle.depth_attachment(o["depth"]) != np.rint(col["depth"].astype(np.float64) * 16777215.0).astype(np.uint32)).sum())}
        if any(diffs.values()):
            bad += 1
            print(f"MISMATCH view {v} seed {a.seed} {w}x{h} limit {int(u[19:20].view(np.uint32)[0])}: {diffs}", flush=True)
        print(f"view {v + 1}/{a.views}: limit {int(u[19:20].view(np.uint32)[0])}, {bad} mismatching so far", flush=True)
    print(f"done: {a.views} views, {bad} mismatching")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
