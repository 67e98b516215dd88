#!/usr/bin/env python3
"""Development aid (GPU box): the exact near-first traversal order against the llvmpipe fixtures of the synthetic scenes --
per fixture and pipeline the pixels whose primary hit slot / distance / colour differ, and the work counters."""
import glob, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import api, scenes

GOLDEN = os.path.join(ROOT, "tests", "golden")
names = sys.argv[1:] or ["100k", "1m"]
for name in names:
    oc = {"100k": scenes.synthetic_100k, "1m": scenes.synthetic_1m}[name](device=0)
    sc = api.Scene.from_dict(oc.as_dict(), device=0)
    refs = oc.m_compressed_triangles
    for f in sorted(glob.glob(os.path.join(GOLDEN, f"refsynth_{name}_*.npz"))):
        z = np.load(f)
        w, h = (int(v) for v in z["size"])
        sc.set_skybox(z["sky"])
        for pipe in ("wavefront", "persistent"):
            for order in ("reference", "near_first_exact"):
                sc.set_pipeline(pipe); sc.set_traversal_order(order)
                sc.set_counting(True)
                g = sc.render(z["uniforms"], w, h)
                st = sc.stats()
                sc.set_counting(False)
                bad_slot = g["slot"] != z["slot"]
                bad_t = g["t"].view(np.uint32) != z["t"].view(np.uint32)
                bad_rgb = (g["rgba"][..., :3].view(np.uint32) != z["rgb"].view(np.uint32)).any(-1)
                print(os.path.basename(f), pipe, order, "slot", int(bad_slot.sum()), "t", int(bad_t.sum()), "rgb", int(bad_rgb.sum()), "kind", int((g["kind"] != z["kind"]).sum()),
                      "| pops", st["node_pops"], "lds", st["node_pops_from_lds"], "fallbacks", st["exact_fallbacks"], "checks", st["exact_candidate_checks"], "entered", st["rays_entering_octree"], "shaded", st["triangle_hits_shaded"])
                g2 = sc.render(z["uniforms"], w, h, planes=False)     # without the slot plane: no candidate checks
                print("      without planes: rgb", int((g2["rgba"][..., :3].view(np.uint32) != z["rgb"].view(np.uint32)).any(-1).sum()), "depth", int((g2["depth"].view(np.uint32) != z["depth"].view(np.uint32)).sum()))
                ys, xs = np.nonzero(bad_slot)
                for y, x in list(zip(ys, xs))[:4]:
                    a, b = int(g["slot"][y, x]), int(z["slot"][y, x])
                    print("   px", y, x, "got slot", a, refs[a].tolist(), "t", g["t"][y, x], "want", b, refs[b].tolist(), "t", z["t"][y, x], "kind", g["kind"][y, x], z["kind"][y, x])
    sc.close()
