"""Model of the mip-mapped lookup against llvmpipe's texture()."""
import os, sys, itertools
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import refrun
from tools.experiments.raster_mip import VERT, frag, words, F
rng = np.random.default_rng(5)
def lerp(a, b, w): return a + (((b - a) * w + 128) >> 8)
def fixed(c, size):
    if size & (size - 1) == 0: return int(np.rint(F(F(c * F(size)) * F(256))))
    f = min(F(c - np.floor(c)), F(0.99999994)); return int(np.rint(F(F(f * F(size)) * F(256))))
def bilinear(img, u, v, repeat=True):
    h, w = img.shape[:2]
    su, sv = fixed(u, w) - 128, fixed(v, h) - 128
    i0, wu, j0, wv = su >> 8, su & 255, sv >> 8, sv & 255
    if repeat: wr = lambda a, n: a % n
    else: wr = lambda a, n: min(max(a, 0), n - 1)
    return lerp(lerp(img[wr(j0, h), wr(i0, w)], img[wr(j0, h), wr(i0 + 1, w)], wu), lerp(img[wr(j0 + 1, h), wr(i0, w)], img[wr(j0 + 1, h), wr(i0 + 1, w)], wu), wv)
def mip_chain(tex):
    levels = [tex.astype(np.int64)]
    while levels[-1].shape[0] > 1 or levels[-1].shape[1] > 1:
        src = levels[-1]; sh, sw = src.shape[:2]; dh, dw = max(sh // 2, 1), max(sw // 2, 1)
        dst = np.zeros((dh, dw, 4), np.int64)
        for j in range(dh):
            for i in range(dw):
                dst[j, i] = bilinear(src, F((i + 0.5) / dw), F((j + 0.5) / dh), repeat=False)
        levels.append(dst)
    return levels
def flog2(x):
    b = np.asarray(x, F).view(np.int32)
    return (b - np.int32(0x3f800000)).astype(F) * F(2.0 ** -23)
def run(TW, TH, uvs, W=32, H=32, seed=0):
    tex = np.random.default_rng(seed).integers(0, 256, (TH, TW, 4), dtype=np.uint8)
    v = np.zeros((3, 8), F); v[:, :3] = [[-1, -1, 0], [3, -1, 0], [-1, 3, 0]]; v[:, 6:8] = uvs
    idx = np.arange(3, dtype=np.uint32)
    t = refrun.run_overlay(v, idx, words, tex, W, H, vert_text=VERT, frag_text=frag("c = texture(texImage, vs_out_tex);"))["rgba"]
    o = refrun.run_overlay(v, idx, words, tex, W, H, vert_text=VERT, frag_text=frag("c = vec4(textureQueryLod(texImage, vs_out_tex), vs_out_tex);"))["rgba"]
    return tex, np.rint(t * 255).astype(np.int64), o[..., 2], o[..., 3], o[..., 1]
def model(tex, u, v, variant):
    H, W = u.shape; TH, TW = tex.shape[:2]
    levels = mip_chain(tex); last = len(levels) - 1
    out = np.zeros((H, W, 4), np.int64)
    for y0 in range(0, H, 2):
        for x0 in range(0, W, 2):
            a = (u[y0, x0 + 1] - u[y0, x0]) * F(TW); c = (v[y0, x0 + 1] - v[y0, x0]) * F(TH)
            b = (u[y0 + 1, x0] - u[y0, x0]) * F(TW); d = (v[y0 + 1, x0] - v[y0, x0]) * F(TH)
            lod = F(flog2(np.maximum(a * a + c * c, b * b + d * d)) * F(0.5))
            for dy in range(2):
                for dx in range(2):
                    uu, vv = u[y0 + dy, x0 + dx], v[y0 + dy, x0 + dx]
                    if lod <= 0: out[y0 + dy, x0 + dx] = bilinear(levels[0], uu, vv); continue
                    l0 = int(np.floor(lod)); fr = F(lod - F(l0))
                    if l0 >= last: out[y0 + dy, x0 + dx] = bilinear(levels[last], uu, vv); continue
                    w8 = {"floor": int(np.floor(fr * F(256))), "rint": int(np.rint(fr * F(256))), "rint255": int(np.rint(fr * F(255)))}[variant]
                    c0, c1 = bilinear(levels[l0], uu, vv), bilinear(levels[l0 + 1], uu, vv)
                    out[y0 + dy, x0 + dx] = lerp(c0, c1, w8)
    return out
if __name__ == "__main__":
    for (TW, TH, uvs) in ((64, 32, [[0.013, 0.21], [4.37, 1.91], [-1.13, 5.77]]), (64, 32, [[0.5, 0.5], [1.9, 0.7], [0.3, 2.2]]), (16, 16, [[0, 0], [9.0, 0.0], [0.0, 9.0]])):
        tex, got, u, v, lod = run(TW, TH, uvs)
        print("texture", TW, TH, "lod range", lod.min(), lod.max())
        for variant in ("floor", "rint", "rint255"):
            m = model(tex, u, v, variant)
            print("   ", variant, "mismatching pixels", int((m != got).any(-1).sum()), "of", got.shape[0] * got.shape[1], "max abs", int(np.abs(m - got).max()))
