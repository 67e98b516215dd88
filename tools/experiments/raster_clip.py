import os, sys, itertools
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import refrun
F = np.float32
FRAG = "#version 430\nin vec3 vs_out_pos; in vec3 vs_out_norm; in vec2 vs_out_tex; out vec4 c;\nvoid main(){ c = vec4(gl_FragCoord.z, gl_FragCoord.w, vs_out_norm.x, 1.0); }\n"
VERT = "#version 430\nlayout(location=0) in vec3 p; layout(location=1) in vec3 n; layout(location=2) in vec2 t; out vec3 vs_out_pos; out vec3 vs_out_norm; out vec2 vs_out_tex; uniform mat4 world; uniform mat4 worldIT; uniform mat4 viewProj;\nvoid main(){ gl_Position = vec4(p, t.x); vs_out_pos = p; vs_out_norm = n; vs_out_tex = vec2(0); }\n"
def fma(a, b, c): return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(F)
I = np.eye(4, dtype=F)
words = refrun.overlay_uniform_words(I, [I], [I], [0, 0, -1], [0.25] * 3, [0.5] * 3)
tex = np.full((4, 4, 4), 255, np.uint8)
W, H = 48, 32
Xf = np.broadcast_to(np.arange(W, dtype=F)[None, :], (H, W)); Yf = np.broadcast_to(np.arange(H, dtype=F)[:, None], (H, W))
def tap(clip, nx):
    v = np.zeros((3, 8), F); v[:, :3] = clip[:, :3]; v[:, 6] = clip[:, 3]; v[:, 3] = nx
    o = refrun.run_overlay(v, np.array([0, 1, 2], np.uint32), words, tex, W, H, frag_text=FRAG, vert_text=VERT)["rgba"]
    return o[..., 3] > 0, o[..., 0], o[..., 1], o[..., 2]
def planes_unclipped(clip):
    oow = F(1.0) / clip[:, 3]; ndc = clip[:, :3] * oow[:, None]
    xw = fma(ndc[:, 0], F(W / 2), F(W / 2)); yw = fma(ndc[:, 1], F(H / 2), F(H / 2)); zw = fma(ndc[:, 2], F(0.5), F(0.5))
    p = [1, 0, 2]; x, y, a = xw[p], yw[p], zw[p]
    dx01, dy01 = x[0] - x[1], y[0] - y[1]; dx20, dy20 = x[2] - x[0], y[2] - y[0]
    ooa = F(1.0) / (dx01 * dy20 - dx20 * dy01)
    da01, da20 = a[0] - a[1], a[2] - a[0]
    dadx = da01 * (dy20 * ooa) - da20 * (dy01 * ooa); dady = da20 * (dx01 * ooa) - da01 * (dx20 * ooa)
    a0 = a[0] - (dadx * (x[0] - F(0.5)) + dady * (y[0] - F(0.5)))
    return fma(dady, Yf, fma(dadx, Xf, a0))
if __name__ == "__main__":
    for out_by in (1.02, 1.2, 1.5, 3.0, 10.0):
        clip = np.ones((3, 4), F); clip[:, :3] = [[-0.6, -0.5, 0.2], [0.7, -0.4, -0.3], [0.1, out_by, 0.6]]
        m, z, w, n = tap(clip, [0.1, 0.5, -0.7])
        zm = planes_unclipped(clip)
        print(f"top vertex at y = {out_by}: covered {m.sum()}, z mismatches against the unclipped plane {(zm.view(np.uint32)[m] != z.view(np.uint32)[m]).sum()}")
    print("--- permutations / clipped-fan hypotheses for top vertex at 1.2")
    clip = np.ones((3, 4), F); clip[:, :3] = [[-0.6, -0.5, 0.2], [0.7, -0.4, -0.3], [0.1, 1.2, 0.6]]
    m, z, w, n = tap(clip, [0.1, 0.5, -0.7])
    def plane_from(win, p):      # win: [k, 3] window x, y, z
        x, y, a = win[p, 0], win[p, 1], win[p, 2]
        dx01, dy01 = x[0] - x[1], y[0] - y[1]; dx20, dy20 = x[2] - x[0], y[2] - y[0]
        ooa = F(1.0) / (dx01 * dy20 - dx20 * dy01)
        da01, da20 = a[0] - a[1], a[2] - a[0]
        dadx = da01 * (dy20 * ooa) - da20 * (dy01 * ooa); dady = da20 * (dx01 * ooa) - da01 * (dx20 * ooa)
        a0 = a[0] - (dadx * (x[0] - F(0.5)) + dady * (y[0] - F(0.5)))
        return fma(dady, Yf, fma(dadx, Xf, a0))
    def window(c, fused=True):
        oow = F(1.0) / c[:, 3]
        if fused: return np.stack([fma(c[:, 0] * oow, F(W / 2), F(W / 2)), fma(c[:, 1] * oow, F(H / 2), F(H / 2)), fma(c[:, 2] * oow, F(0.5), F(0.5))], 1)
        return np.stack([c[:, 0] * oow * F(W / 2) + F(W / 2), c[:, 1] * oow * F(H / 2) + F(H / 2), c[:, 2] * oow * F(0.5) + F(0.5)], 1)
    win = window(clip)
    for p in itertools.permutations(range(3)):
        zm = plane_from(win, list(p)); print(p, int((zm.view(np.uint32)[m] != z.view(np.uint32)[m]).sum()))
    # clip against y <= w: plane dist = w - y
    def dist(c): return c[3] - c[1]
    poly = []
    vs = [clip[0], clip[1], clip[2]]
    for variant in range(4):
        out = []
        prev = vs[-1]; dpp = dist(prev)
        for cur in vs:
            dp = dist(cur)
            if dpp >= 0: out.append(("orig", prev))
            if (dp < 0) != (dpp < 0):
                if dp < 0:
                    t = dp / (dp - dpp); new = cur + t * (prev - cur) if variant % 2 == 0 else fma(np.full(4, t, F), prev - cur, cur)
                else:
                    t = dpp / (dpp - dp); new = prev + t * (cur - prev) if variant % 2 == 0 else fma(np.full(4, t, F), cur - prev, prev)
                out.append(("new", new.astype(F)))
            prev, dpp = cur, dp
        pts = np.array([p for _, p in out], F)
        kinds = [k for k, _ in out]
        winp = np.zeros((len(pts), 3), F)
        for i, (k, p) in enumerate(out):
            winp[i] = window(p[None], fused=(k == "orig" or variant >= 2))[0]
        print("variant", variant, "polygon", kinds)
        # each pixel must match one of the fan triangles' planes: count pixels matching none, over fan roots and vertex orders
        for root in range(len(pts)):
            idx = [(root + i) % len(pts) for i in range(len(pts))]
            best = None
            for order in itertools.permutations(range(3)):
                ok = np.zeros((H, W), bool)
                for i in range(1, len(pts) - 1):
                    tri = [idx[0], idx[i], idx[i + 1]]
                    zm = plane_from(winp, [tri[o] for o in order])
                    ok |= zm.view(np.uint32) == z.view(np.uint32)
                bad = int((~ok & m).sum())
                if best is None or bad < best[0]: best = (bad, order)
            print("   fan root", root, "unmatched pixels", best)
