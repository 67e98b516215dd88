#!/usr/bin/env python3
"""Times the host octree builder against the device builder on the bench scenes and checks that the arrays agree; and the way from
meshes to a device-resident scene in two steps (builder, rt_scene_create) against rt_scene_create_from_meshes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import api, scenes

cases = {
    "bunny (69k triangles, depth 18)": (lambda: [scenes.bunny_mesh()], (18, 10, 6, 6)),
    "synthetic 100k primitives (1.6M triangles, depth 8)": (lambda: scenes.synthetic_primitives(100_000, seed=1), (8, 10, 6, 6)),
    "synthetic 1M (1.28M triangles, depth 18)": (lambda: scenes.synthetic_primitives(1000, seed=2, sphere_subdiv=3, extent=3.0, size_range=(0.15, 0.4), materials=(1, 5, 6, 2), perturb=0.01, boxes=False), (18, 10, 6, 6)),
}
api.Octree([scenes.bunny_mesh()], 4, 10, 6, 6, device=0)     # first use of the device: runtime start-up
for name, (make, params) in cases.items():
    meshes = make()
    t0 = time.perf_counter(); host = api.Octree(meshes, *params); t1 = time.perf_counter()
    dev = api.Octree(meshes, *params, device=0); t2 = time.perf_counter()
    dev = api.Octree(meshes, *params, device=0); t3 = time.perf_counter()
    t4 = time.perf_counter(); two = api.Scene.from_octree(dev, device=0); t5 = time.perf_counter(); two.close()
    one = api.Scene.from_meshes(meshes, *params, device=0); t6 = time.perf_counter(); one.close()
    t6b = time.perf_counter(); one = api.Scene.from_meshes(meshes, *params, device=0); t7 = time.perf_counter(); one.close()
    print(f"{name}: to a scene on the device: build + create {t3 - t2 + t5 - t4:.3f} s, in one call without leaving the device {t7 - t6b:.3f} s", flush=True)
    ok = all(np.array_equal(host.as_dict()[k].view(np.uint32), dev.as_dict()[k].view(np.uint32)) for k in host.as_dict())
    print(f"{name}: host {t1 - t0:.3f} s, device {t3 - t2:.3f} s (first call {t2 - t1:.3f} s), identical={ok}, refs={len(host.m_compressed_triangles)}, words={len(host.m_compressed_node_buffer)}", flush=True)
