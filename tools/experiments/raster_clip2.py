import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import refrun, raster
F = np.float32
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
tex = rng.integers(0, 256, (16, 16, 4), dtype=np.uint8); tex[..., 3] = 255
I = np.eye(4, dtype=F)
words = raster.overlay_words(I.reshape(-1), [I.reshape(-1)], [I.reshape(-1)], (-1, -2, -0.5), (0.1, 0.2, 0.3), (0.9, 0.8, 0.7))
W, H = 96, 64
stats = {}
n = 0
while n < 60:
    v = np.zeros((3, 8), F)
    v[:, :3] = rng.uniform(-1.0, 1.0, (1, 3)) + rng.uniform(-1.3, 1.3, (3, 3)); v[:, 3:6] = rng.uniform(-1, 1, (3, 3))
    masks = [tuple(int(b) for b in (p[0] > 1, p[0] < -1, p[1] > 1, p[1] < -1, p[2] < -1, p[2] > 1)) for p in v[:, :3]]
    any_out = tuple(int(any(m[i] for m in masks)) for i in range(6))
    if sum(any_out) == 0: continue
    o = refrun.run_overlay(v, np.arange(3, dtype=np.uint32), words, tex, W, H)
    ref24 = np.rint(o["depth"].astype(np.float64) * 16777215.0).astype(np.uint32)
    if (ref24 != 0xFFFFFF).sum() < 10: continue
    n += 1
    col = np.zeros((H, W, 4), F); d24 = np.full((H, W), 0xFFFFFF, np.uint32)
    raster.draw(v, np.arange(3, dtype=np.uint32), words, tex, col, d24)
    bad = int((d24 != ref24).sum())
    key = "".join(str(b) for b in any_out) + " verts-out:" + "".join(str(int(any(m))) for m in masks)
    s = stats.setdefault(key, [0, 0, 0]); s[0] += 1; s[1] += int(bad > 0); s[2] += bad
for k in sorted(stats): print(k, "triangles", stats[k][0], "with depth mismatches", stats[k][1], "pixels", stats[k][2])
