import itertools, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import refrun
F = np.float32
FRAG = "#version 430\nin vec3 vs_out_pos; in vec3 vs_out_norm; in vec2 vs_out_tex; out vec4 c;\nvoid main(){ c = vec4(vs_out_norm.x, vs_out_norm.y, vs_out_norm.z, 1.0); }\n"
VERT = "#version 430\nlayout(location=0) in vec3 p; layout(location=1) in vec3 n; layout(location=2) in vec2 t; out vec3 vs_out_pos; out vec3 vs_out_norm; out vec2 vs_out_tex; uniform mat4 world; uniform mat4 worldIT; uniform mat4 viewProj;\nvoid main(){ gl_Position = vec4(p, t.x); vs_out_pos = p; vs_out_norm = n; vs_out_tex = vec2(0); }\n"
def fma(a, b, c): return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(F)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
I = np.eye(4, dtype=F)
words = refrun.overlay_uniform_words(I, [I], [I], [0, 0, -1], [0.25] * 3, [0.5] * 3)
tex = np.full((4, 4, 4), 255, np.uint8)
W, H = 48, 32
Xf = np.broadcast_to(np.arange(W, dtype=F)[None, :], (H, W)); Yf = np.broadcast_to(np.arange(H, dtype=F)[:, None], (H, W))
score = {}
n_done = 0
while n_done < 24:
    clip = np.ones((3, 4), F)
    clip[:, :3] = rng.uniform(-1, 1, (3, 3)).astype(F)
    if n_done % 2:
        clip[:, 3] = rng.uniform(0.5, 3.0, 3).astype(F); clip[:, :3] *= clip[:, 3:4]
    v = np.zeros((3, 8), F); v[:, :3] = clip[:, :3]; v[:, 6] = clip[:, 3]; v[:, 3:6] = rng.uniform(-1, 1, (3, 3))
    o = refrun.run_overlay(v, np.array([0, 1, 2], np.uint32), words, tex, W, H, frag_text=FRAG, vert_text=VERT)
    m = o["rgba"][..., 3] > 0
    if m.sum() < 20: continue
    n_done += 1
    oow = F(1.0) / clip[:, 3]; ndc = clip[:, :3] * oow[:, None]
    xw = fma(ndc[:, 0], F(W / 2), F(W / 2)); yw = fma(ndc[:, 1], F(H / 2), F(H / 2))
    p = [1, 0, 2]
    x, y = xw[p], yw[p]
    dx01, dy01 = x[0] - x[1], y[0] - y[1]; dx20, dy20 = x[2] - x[0], y[2] - y[0]
    ooa = F(1.0) / (dx01 * dy20 - dx20 * dy01)
    def plane(a):
        da01, da20 = a[0] - a[1], a[2] - a[0]
        dadx = da01 * (dy20 * ooa) - da20 * (dy01 * ooa); dady = da20 * (dx01 * ooa) - da01 * (dx20 * ooa)
        a0 = a[0] - (dadx * (x[0] - F(0.5)) + dady * (y[0] - F(0.5)))
        return fma(dady, Yf, fma(dadx, Xf, a0))
    Wm = plane(oow[p])
    for c in range(3):
        n = v[:, 3 + c][p]
        cands = {"a*oow, A*(1/W)": plane(n * oow[p]) * (F(1.0) / Wm), "a*oow, A/W": plane(n * oow[p]) / Wm, "a/w, A*(1/W)": plane(n / clip[:, 3][p]) * (F(1.0) / Wm), "a/w, A/W": plane(n / clip[:, 3][p]) / Wm}
        for k, am in cands.items():
            score[k] = score.get(k, 0) + int((am.view(np.uint32)[m] != o["rgba"][..., c].view(np.uint32)[m]).sum())
    score["pixels"] = score.get("pixels", 0) + 3 * int(m.sum())
print(score)
