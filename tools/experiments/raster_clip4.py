import os, sys, itertools
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.experiments.raster_clip import tap, fma, F, W, H, Xf, Yf
rng = np.random.default_rng(1)
clips = []
for trial in range(9):
    clip = np.ones((3, 4), F); clip[:, :3] = rng.uniform(-0.9, 0.9, (3, 3))
    which = trial % 3
    clip[which, 0] = rng.uniform(1.1, 2.0)
    if trial >= 6: clip[(which + 1) % 3, 0] = rng.uniform(1.1, 2.0)
    a = (clip[1, 0] - clip[0, 0]) * (clip[2, 1] - clip[0, 1]) - (clip[2, 0] - clip[0, 0]) * (clip[1, 1] - clip[0, 1])
    if a < 0: clip[:, 1] = -clip[:, 1]
    clips.append(clip)
clip = clips[2]
print(clip)
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
    oow = F(1.0) / c[3]
    if fused: return np.array([fma(c[0] * oow, F(W / 2), F(W / 2)), fma(c[1] * oow, F(H / 2), F(H / 2)), fma(c[2] * oow, F(0.5), F(0.5))], F)
    return np.array([c[0] * oow * F(W / 2) + F(W / 2), c[1] * oow * F(H / 2) + F(H / 2), c[2] * oow * F(0.5) + F(0.5)], F)
v0, v1, v2 = clip
d = lambda c: c[3] - c[0]
# polygon: v0, v1, N12 (going out), N20 (coming in)
t12 = d(v2) / (d(v2) - d(v1)); N12 = v2 + t12 * (v1 - v2)
t20 = d(v2) / (d(v2) - d(v0)); N20 = v2 + t20 * (v0 - v2)
poly = [("o", v0), ("o", v1), ("n", N12.astype(F)), ("n", N20.astype(F))]
win = np.array([window(p, k == "o") for k, p in poly], F)
print(win)
tris = [(1, 2, 0), (2, 3, 0)]
for ti, tri in enumerate(tris):
    for order in ((0, 1, 2), (1, 0, 2), (2, 0, 1)):
        zm = plane_from(win, [tri[o] for o in order])
        eq = (zm.view(np.uint32) == z.view(np.uint32)) & m
        print("fan tri", ti, tri, "origin", tri[order[0]], "matches", int(eq.sum()), "of", int(m.sum()))
zm1 = plane_from(win, [2, 1, 0]); zm2 = plane_from(win, [3, 2, 0])
ok = (zm1.view(np.uint32) == z.view(np.uint32)) | (zm2.view(np.uint32) == z.view(np.uint32))
for r in range(H - 1, -1, -1):
    print("".join("." if not m[r, c] else ("1" if zm1.view(np.uint32)[r, c] == z.view(np.uint32)[r, c] else ("2" if zm2.view(np.uint32)[r, c] == z.view(np.uint32)[r, c] else "X")) for c in range(W)))
print("--- tri 0 variants")
for f0, f1, f12 in itertools.product((True, False), repeat=3):
    wv = np.array([window(v0, f0), window(v1, f1), window(N12.astype(F), f12)], F)
    for order in ((0, 1, 2), (1, 0, 2), (2, 0, 1)):
        zm = plane_from(wv, list(order))
        eq = (zm.view(np.uint32) == z.view(np.uint32)) & m
        print("fused v0", f0, "v1", f1, "N12", f12, "origin", ["v0", "v1", "N12"][order[0]], "matches", int(eq.sum()))
print("--- tri 0 per-origin maps")
wv = np.array([window(v0, True), window(v1, True), window(N12.astype(F), False)], F)
zs = [plane_from(wv, list(order)) for order in ((0, 1, 2), (1, 0, 2), (2, 0, 1))]
tri0 = m & ~((zm2.view(np.uint32) == z.view(np.uint32)))
for r in range(H - 1, -1, -1):
    row = ""
    for c in range(W):
        if not tri0[r, c]: row += "."; continue
        ds = [int(zz.view(np.int32)[r, c]) - int(z.view(np.int32)[r, c]) for zz in zs]
        row += "[" + ",".join("%+d" % d for d in ds) + "]"
    if "[" in row: print(r, row.replace(".", ""))
print("--- N12 variants")
dv1, dv2 = d(v1), d(v2)
cands = {}
t_a = dv2 / (dv2 - dv1); t_b = dv1 / (dv1 - dv2)
cands["out + t_a (in - out)"] = v2 + t_a * (v1 - v2)
cands["in + t_b (out - in)"] = v1 + t_b * (v2 - v1)
cands["out*(1-t_b) ..."] = v1 * (F(1) - t_b) + v2 * t_b
cands["fma out + t_a(in-out)"] = fma(np.full(4, t_a, F), v1 - v2, v2)
cands["fma in + t_b(out-in)"] = fma(np.full(4, t_b, F), v2 - v1, v1)
cands["in + (1-t_a)(out-in)"] = v1 + (F(1) - t_a) * (v2 - v1)
for name, N in cands.items():
    N = N.astype(F)
    for fw in (True, False):
        wv = np.array([window(v0, True), window(v1, True), window(N, fw), window(N20.astype(F), False)], F)
        z1 = plane_from(wv, [2, 1, 0]); z2 = plane_from(wv, [3, 2, 0])
        ok = ((z1.view(np.uint32) == z.view(np.uint32)) | (z2.view(np.uint32) == z.view(np.uint32)))
        print(f"{name:28s} fused window {fw}: unmatched {int((~ok & m).sum())}   N = {N}")
