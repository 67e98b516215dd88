#!/usr/bin/env python3
"""Development aid (GPU box): many 4K frames of a scene rendered in the shader's child order and in the exact near-first order (wavefront
pipeline, device targets), compared on the device: pixels whose colour or depth bits differ, per frame and in total.  The frames differ
in time (sub-pixel jitter, scattering) and camera pose, so the count is over billions of distinct rays.
usage: tools/compare_orders_4k.py <bunny|s100k|s1m> [frames] [limit] [wavefront|persistent]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracer_amd import api, scenes
from raytracer_amd.camera import Camera
from raytracer_amd.params import Uniforms

name = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 100
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 5
pipeline = sys.argv[4] if len(sys.argv) > 4 else "wavefront"
W, H = 3840, 2160
oc = {"bunny": scenes.default_octree, "s100k": scenes.synthetic_100k, "s1m": scenes.synthetic_1m}[name](device=0)
sc = api.Scene.from_dict(oc.as_dict(), device=0)
sc.set_skybox(scenes.procedural_skybox(900))
sc.set_pipeline(pipeline)
stream = torch.cuda.current_stream().cuda_stream
bufs = {o: (torch.zeros((H, W, 4), device="cuda"), torch.zeros((H, W), device="cuda")) for o in ("reference", "near_first_exact")}
rng = np.random.default_rng(7)
total = rays = 0
for k in range(frames):
    if k % 10 == 0:      # a new pose every ten frames: around the scene, sometimes inside it
        a, r = rng.uniform(0, 2 * np.pi), rng.uniform(0.3, 4.0)
        cam = Camera(eye=(float(r * np.cos(a)), float(rng.uniform(0.3, 3.0)), float(r * np.sin(a))), at=(float(rng.uniform(-0.5, 0.5)), float(rng.uniform(0.5, 1.5)), float(rng.uniform(-0.5, 0.5))), aspect=W / H) if k else Camera(aspect=W / H)
    u = Uniforms.from_camera(cam, time=k / 60.0, max_recursion_limit=limit).pack()
    for order, (col, dep) in bufs.items():
        sc.set_traversal_order(order)
        tg = api.rt_target()
        tg.memory, tg.sync, tg.color, tg.depth = api.RT_MEM_DEVICE, 1, col.data_ptr(), dep.data_ptr()
        sc.render_raw(api.make_frame(u, W, H, accum_index=1, blend=False, stream=stream), tg)
        assert sc.last_mode() == (pipeline, order), sc.last_mode()
    (ca, da), (cb, db) = bufs["reference"], bufs["near_first_exact"]
    bad = (ca.view(torch.int32) != cb.view(torch.int32)).any(-1) | (da.view(torch.int32) != db.view(torch.int32))
    n = int(bad.sum())
    total += n
    if n:
        ys, xs = torch.nonzero(bad, as_tuple=True)
        print(f"frame {k}: {n} differing pixel(s), first at (y, x) = ({int(ys[0])}, {int(xs[0])}); depth {float(da[ys[0], xs[0]])!r} vs {float(db[ys[0], xs[0]])!r}", flush=True)
    if k % 20 == 19:
        print(f"{k + 1} frames, {total} differing pixels so far", flush=True)
print(f"done: {name}, {pipeline}, limit {limit}: {frames} frames of {W}x{H}, {total} differing pixels")
sc.close()
