import os, sys, itertools
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.experiments.raster_clip import tap, fma, F, W, H, Xf, Yf
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 2)
def plane_from(win, p):
    x, y, a = win[p, 0], win[p, 1], win[p, 2]
    dx01, dy01 = x[0] - x[1], y[0] - y[1]; dx20, dy20 = x[2] - x[0], y[2] - y[0]
    ooa = F(1.0) / (dx01 * dy20 - dx20 * dy01)
    da01, da20 = a[0] - a[1], a[2] - a[0]
    dadx = da01 * (dy20 * ooa) - da20 * (dy01 * ooa); dady = da20 * (dx01 * ooa) - da01 * (dx20 * ooa)
    a0 = a[0] - (dadx * (x[0] - F(0.5)) + dady * (y[0] - F(0.5)))
    return fma(dady, Yf, fma(dadx, Xf, a0))
def window(c, fused):
    oow = F(1.0) / c[3]
    if fused: return np.array([fma(c[0] * oow, F(W / 2), F(W / 2)), fma(c[1] * oow, F(H / 2), F(H / 2)), fma(c[2] * oow, F(0.5), F(0.5))], F)
    return np.array([c[0] * oow * F(W / 2) + F(W / 2), c[1] * oow * F(H / 2) + F(H / 2), c[2] * oow * F(0.5) + F(0.5)], F)
PLANES = [lambda c: c[3] - c[0], lambda c: c[3] + c[0], lambda c: c[3] - c[1], lambda c: c[3] + c[1], lambda c: c[3] + c[2], lambda c: c[3] - c[2]]
def newv(a, b, da, db, based):      # edge a -> b, one of them outside
    if based == "a": t = da / (da - db); return (a + t * (b - a)).astype(F)
    t = db / (db - da); return (b + t * (a - b)).astype(F)
n = 0
while n < 40:
    clip = np.ones((3, 4), F); clip[:, :3] = rng.uniform(-0.9, 0.9, (3, 3))
    pl = int(rng.integers(0, 6)); which = int(rng.integers(0, 3)); two = rng.uniform() < 0.4
    axis, sign = [(0, 1), (0, -1), (1, 1), (1, -1), (2, -1), (2, 1)][pl]
    clip[which, axis] = sign * rng.uniform(1.1, 2.0)
    if two: clip[(which + 1) % 3, axis] = sign * rng.uniform(1.1, 2.0)
    a = (clip[1, 0] - clip[0, 0]) * (clip[2, 1] - clip[0, 1]) - (clip[2, 0] - clip[0, 0]) * (clip[1, 1] - clip[0, 1])
    if a < 0:
        if axis == 1: clip[:, 0] = -clip[:, 0]
        else: clip[:, 1] = -clip[:, 1]
    m, z, w, nn = tap(clip, [0.1, 0.5, -0.7])
    if m.sum() < 30: continue
    n += 1
    dist = PLANES[pl]
    vs = [clip[0], clip[1], clip[2]]
    # walk from v0
    edges = []      # polygon entries: ("o", vertex) or ("n", (a, b, da, db, direction))
    prev = vs[0]; dpp = dist(prev)
    for i in range(1, 4):
        cur = vs[i % 3]; dp = dist(cur)
        if dpp >= 0: edges.append(("o", prev))
        if (dp < 0) != (dpp < 0): edges.append(("n", (prev, cur, dpp, dp, "out" if dp < 0 else "in")))
        prev, dpp = cur, dp
    news = [i for i, e in enumerate(edges) if e[0] == "n"]
    results = {}
    for combo in itertools.product("ab", repeat=len(news)):
        pts = []
        for i, e in enumerate(edges):
            if e[0] == "o": pts.append(window(e[1], True))
            else:
                a_, b_, da, db, _ = e[1]; pts.append(window(newv(a_, b_, da, db, combo[news.index(i)]), False))
        pts = np.array(pts, F)
        ok = np.zeros((H, W), bool)
        for q in range(2, len(pts)):
            zm = plane_from(pts, [q, q - 1, 0]); ok |= zm.view(np.uint32) == z.view(np.uint32)
        results[combo] = int((~ok & m).sum())
    dirs = [(edges[i][1][4], "t_a=%.2f" % (edges[i][1][2] / (edges[i][1][2] - edges[i][1][3]))) for i in news]
    good = [c for c, r in results.items() if r == 0]
    print("plane", pl, "vertex", which, "two" if two else "one", "new-vertex directions", dirs, "px", int(m.sum()), "ok:", good if good else results)
