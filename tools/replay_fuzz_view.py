#!/usr/bin/env python3
"""Development aid (GPU box): replays one view of tests/fuzz_parity.py (same seed, same random stream) and prints, for the
pixels where the exact near-first order rendered WITHOUT the slot plane differs from the reference order, both primary hits.
usage: tools/replay_fuzz_view.py <scene> <seed> <view>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import fuzz_parity
from raytracer_amd import api, scenes

scene, seed, view = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(seed)
oc = {"bunny": scenes.default_octree, "small": scenes.small_test_scene, "s100k": scenes.synthetic_100k, "s1m": scenes.synthetic_1m, "deep": scenes.deep_test_scene, "planes": scenes.plane_aligned_scene}[scene]()
sky = rng.integers(0, 256, size=(6, 48, 48, 4), dtype=np.uint8)
for v in range(view + 1):
    u, w, h = fuzz_parity.random_view(rng, lattice=scene == "planes")
np.save(os.path.join(ROOT, "gpurun_out", f"replay_{scene}_{seed}_{view}_uniforms.npy"), u)
print("view", view, w, h, "limit", int(u[19:20].view(np.uint32)[0]))
sc = api.Scene.from_octree(oc); sc.set_skybox(sky)
sc.set_pipeline("wavefront")
ref = sc.render(u, w, h)
sc.set_traversal_order("near_first_exact")

def render_without_slot():
    out = {k: np.zeros((h, w), dt) for k, dt in (("depth", np.float32), ("kind", np.uint32), ("t", np.float32), ("nrays", np.uint32))}
    out["rgba"] = np.zeros((h, w, 4), np.float32)
    fr = api.make_frame(u, w, h)
    tg = api.rt_target()
    tg.memory, tg.sync = api.RT_MEM_HOST, 1
    tg.color, tg.depth, tg.hit_kind, tg.hit_t, tg.n_rays = (out[k].ctypes.data for k in ("rgba", "depth", "kind", "t", "nrays"))
    sc.render_raw(fr, tg)
    return out

got = render_without_slot()
bad = (got["rgba"].view(np.uint32) != ref["rgba"].view(np.uint32)).any(-1) | (got["depth"].view(np.uint32) != ref["depth"].view(np.uint32))
for y, x in zip(*np.nonzero(bad)):
    print("pixel", y, x, "reference: kind", ref["kind"][y, x], "slot", ref["slot"][y, x], "t", repr(float(ref["t"][y, x])), "nrays", ref["nrays"][y, x], "depth", repr(float(ref["depth"][y, x])),
          "| near-first: kind", got["kind"][y, x], "t", repr(float(got["t"][y, x])), "nrays", got["nrays"][y, x], "depth", repr(float(got["depth"][y, x])))
    print("   rgba ref", ref["rgba"][y, x], "got", got["rgba"][y, x])
