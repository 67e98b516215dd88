import os, sys, itertools
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.experiments.raster_clip import *
rng = np.random.default_rng(1)
def analyse(clip):
    m, z, w, n = tap(clip, [0.1, 0.5, -0.7])
    def plane_from(win, p):
        x, y, a = win[p, 0], win[p, 1], win[p, 2]
        dx01, dy01 = x[0] - x[1], y[0] - y[1]; dx20, dy20 = x[2] - x[0], y[2] - y[0]
        ooa = F(1.0) / (dx01 * dy20 - dx20 * dy01)
        da01, da20 = a[0] - a[1], a[2] - a[0]
        dadx = da01 * (dy20 * ooa) - da20 * (dy01 * ooa); dady = da20 * (dx01 * ooa) - da01 * (dx20 * ooa)
        a0 = a[0] - (dadx * (x[0] - F(0.5)) + dady * (y[0] - F(0.5)))
        return fma(dady, Yf, fma(dadx, Xf, a0))
    def window(c, fused):
        oow = F(1.0) / c[:, 3]
        if fused: return np.stack([fma(c[:, 0] * oow, F(W / 2), F(W / 2)), fma(c[:, 1] * oow, F(H / 2), F(H / 2)), fma(c[:, 2] * oow, F(0.5), F(0.5))], 1)
        return np.stack([c[:, 0] * oow * F(W / 2) + F(W / 2), c[:, 1] * oow * F(H / 2) + F(H / 2), c[:, 2] * oow * F(0.5) + F(0.5)], 1)
    def dist(c): return c[3] - c[0]      # x <= w
    vs = [clip[0], clip[1], clip[2]]
    res = {}
    for start, lerpv, winv in itertools.product(range(3), range(2), range(2)):
        order = [vs[(start + i) % 3] for i in range(3)]
        out = []
        prev = order[0]; dpp = dist(prev)
        for i in range(1, 4):
            cur = order[i % 3]; dp = dist(cur)
            if dpp >= 0: out.append(("orig", prev))
            if (dp < 0) != (dpp < 0):
                if dp < 0:
                    t = dp / (dp - dpp); new = cur + t * (prev - cur) if lerpv == 0 else fma(np.full(4, t, F), prev - cur, cur)
                else:
                    t = dpp / (dpp - dp); new = prev + t * (cur - prev) if lerpv == 0 else fma(np.full(4, t, F), cur - prev, prev)
                out.append(("new", new.astype(F)))
            prev, dpp = cur, dp
        winp = np.array([window(p[None], fused=(k == "orig" or winv == 1))[0] for k, p in out], F)
        npts = len(out)
        for root in range(npts):
            idx = [(root + i) % npts for i in range(npts)]
            for order3 in ((0, 1, 2), (1, 0, 2), (2, 0, 1)):
                ok = np.zeros((H, W), bool)
                for i in range(1, npts - 1):
                    tri = [idx[0], idx[i], idx[i + 1]]
                    zm = plane_from(winp, [tri[o] for o in order3])
                    ok |= zm.view(np.uint32) == z.view(np.uint32)
                res[(start, lerpv, winv, root, order3)] = int((~ok & m).sum())
    best = sorted(res.items(), key=lambda kv: kv[1])[:4]
    print("covered", int(m.sum()), "kinds", [k for k, _ in out], "best:", best)
for trial in range(9):
    clip = np.ones((3, 4), F); clip[:, :3] = rng.uniform(-0.9, 0.9, (3, 3))
    which = trial % 3
    clip[which, 0] = rng.uniform(1.1, 2.0)      # one vertex beyond x = w
    if trial >= 6: clip[(which + 1) % 3, 0] = rng.uniform(1.1, 2.0)
    a = (clip[1, 0] - clip[0, 0]) * (clip[2, 1] - clip[0, 1]) - (clip[2, 0] - clip[0, 0]) * (clip[1, 1] - clip[0, 1])
    if a < 0:
        clip[:, 1] = -clip[:, 1]
    print("vertex", which, "out" if trial < 6 else "and next out", end=": ")
    analyse(clip)
