#!/usr/bin/env python3
"""Per-wave lifetime histogram of the persistent path kernel (needs a library built with -DPATH_TRACE)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import api, scenes
from raytracer_amd.camera import Camera
from raytracer_amd.params import Uniforms
lim = int(sys.argv[1]) if len(sys.argv) > 1 else 2
W, H = 3840, 2160
sc = api.Scene.from_dict(scenes.default_octree().as_dict()); sc.set_skybox(scenes.procedural_skybox(900)); sc.set_pipeline("persistent"); sc.set_subframes(1)
color = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
for k in range(3):
    fr = api.make_frame(Uniforms.from_camera(Camera(aspect=W / H), time=k / 60.0, max_recursion_limit=lim).pack(), W, H, accum_index=k + 1)
    tg = api.rt_target(); tg.memory, tg.sync, tg.color = api.RT_MEM_DEVICE, 1, color.data_ptr()
    sc.render_raw(fr, tg)
n = 64 * 64 * 4 + 16384 * 16
buf = np.zeros(n // 8, np.uint64)
L = api.lib(); L.rt_internal_read_pool.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
assert L.rt_internal_read_pool(sc._h, buf.ctypes.data, n) == 0
tr = buf[64 * 64 * 4 // 8:].reshape(-1, 2)
tr = tr[tr[:, 1] > 0].astype(np.int64)
t0 = tr[:, 0].min()
start, end = (tr[:, 0] - t0) / 100.0, (tr[:, 1] - t0) / 100.0   # wall clock ticks at 100 MHz -> microseconds
print("waves", len(tr), "start max %.1f us" % start.max(), "end min/median/max %.1f %.1f %.1f us" % (end.min(), np.median(end), end.max()))
for q in (10, 25, 50, 75, 90, 95, 99, 100):
    print("  %3d%% of waves ended by %.1f us" % (q, np.percentile(end, q)))
edges = np.linspace(0, end.max(), 21)
alive = [(int(((start <= t) & (end > t)).sum())) for t in edges]
print("alive waves at 5% steps:", alive)
