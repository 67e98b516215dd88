"""Texture coordinate to fixed point at ties: a grid of small triangles, each with its own constant texture coordinate."""
import os, sys, itertools
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import refrun
F = np.float32
VERT = "#version 430\nlayout(location=0) in vec3 p; layout(location=1) in vec3 n; layout(location=2) in vec2 t; out vec3 vs_out_pos; out vec3 vs_out_norm; out vec2 vs_out_tex; uniform mat4 world; uniform mat4 worldIT; uniform mat4 viewProj;\nvoid main(){ gl_Position = vec4(p, 1); vs_out_pos = p; vs_out_norm = n; vs_out_tex = t; }\n"
FRAG = "#version 430\nin vec3 vs_out_pos; in vec3 vs_out_norm; in vec2 vs_out_tex; out vec4 c; layout(binding = 0) uniform sampler2D texImage;\nvoid main(){ c = texture(texImage, vs_out_tex); }\n"
rng = np.random.default_rng(2)
TW, TH = 5, 7
tex = rng.integers(0, 256, (TH, TW, 4), dtype=np.uint8)
I = np.eye(4, dtype=F)
words = refrun.overlay_uniform_words(I, [I], [I], [0, 0, -1], [0.25] * 3, [0.5] * 3)
G = 16; W = H = 128
verts = []; uvs = []
for gy in range(G):
    for gx in range(G):
        x0, y0 = -1 + 2 * gx / G, -1 + 2 * gy / G; d = 2 / G * 0.9
        m = int(rng.integers(-3000, 3000))
        kind = (gx + gy) % 4
        target = m + 0.5 + (0, 1e-4 * abs(m), -1e-4 * abs(m), 0.25)[kind] * (1 if kind != 3 else 1)
        u = F(target / (TW * 256))
        v = F(rng.integers(-2000, 2000) / (TH * 256))       # exact multiples: weight and texel unambiguous... (v*TH*256 is an integer up to rounding)
        for px, py in ((x0, y0), (x0 + d, y0), (x0, y0 + d)):
            verts.append([px, py, 0, 0, 0, 1, u, v])
        uvs.append((u, v, gx, gy))
verts = np.array(verts, F)
o = refrun.run_overlay(verts, np.arange(len(verts), dtype=np.uint32), words, tex, W, H, frag_text=FRAG, vert_text=VERT)["rgba"]
got = np.rint(o * 255).astype(np.int64)
t = tex.astype(np.int64)
def lerp(a, b, w): return a + (((b - a) * w + 128) >> 8)
def sample(su, sv):
    su -= 128; sv -= 128
    i0, wu, j0, wv = su >> 8, su & 255, sv >> 8, sv & 255
    i1, j1 = (i0 + 1) % TW, (j0 + 1) % TH; i0 %= TW; j0 %= TH
    return lerp(lerp(t[j0, i0], t[j0, i1], wu), lerp(t[j1, i0], t[j1, i1], wu), wv)
rules = {"rint (half even)": lambda x: int(np.rint(x)), "floor(x + 0.5)": lambda x: int(np.floor(F(x + F(0.5)))), "trunc(x + 0.5)": lambda x: int(np.trunc(F(x + F(0.5)))),
         "half away": lambda x: int(np.sign(x) * np.floor(abs(x) + 0.5)), "floor": lambda x: int(np.floor(x)), "half odd": lambda x: (int(np.floor(x)) | 1) if (x - np.floor(x)) == 0.5 else int(np.rint(x)), "ceil(x - 0.5)": lambda x: int(np.ceil(F(x - F(0.5))))}
score = {k: 0 for k in rules}; n = 0
alt = {"fract first, (f*size)*256": 0, "fract first, f*(size*256)": 0, "rint(exact u*size*256)": 0, "rint(u*(size*256))": 0, "rint(fma(u, size*256, -128))+128": 0, "rint((u*size - 0.5)*256)+128": 0}
def fma(a, b, c): return F(np.float64(a) * np.float64(b) + np.float64(c))
for (u, v, gx, gy) in uvs:
    px, py = int((gx + 0.25) * W / G), int((gy + 0.25) * H / G)
    if o[py, px, 3] == 0 and not o[py, px].any(): continue
    n += 1
    xu = F(F(u * F(TW)) * F(256)); xv = F(F(v * F(TH)) * F(256))
    for k, f in rules.items():
        if not np.array_equal(sample(f(xu), f(xv)), got[py, px]): score[k] += 1
    a1 = (int(np.rint(F(u * F(TW * 256)))), int(np.rint(F(v * F(TH * 256)))))
    a2 = (int(np.rint(fma(u, F(TW * 256), F(-128)))) + 128, int(np.rint(fma(v, F(TH * 256), F(-128)))) + 128)
    a3 = (int(np.rint(F(F(F(u * F(TW)) - F(0.5)) * F(256)))) + 128, int(np.rint(F(F(F(v * F(TH)) - F(0.5)) * F(256)))) + 128)
    a0 = (int(np.rint(np.float64(u) * (TW * 256))), int(np.rint(np.float64(v) * (TH * 256))))
    fu, fv = F(u - np.floor(u)), F(v - np.floor(v))
    b0 = (int(np.rint(F(F(fu * F(TW)) * F(256)))), int(np.rint(F(F(fv * F(TH)) * F(256)))))
    b1 = (int(np.rint(F(fu * F(TW * 256)))), int(np.rint(F(fv * F(TH * 256)))))
    for name, (cu, cv) in zip(alt, (b0, b1, a0, a1, a2, a3)):
        if not np.array_equal(sample(cu, cv), got[py, px]): alt[name] += 1
print(n, score, alt)
for (u, v, gx, gy) in uvs:
    px, py = int((gx + 0.25) * W / G), int((gy + 0.25) * H / G)
    if not o[py, px].any(): continue
    xu = F(F(u * F(TW)) * F(256)); xv = F(F(v * F(TH)) * F(256))
    if not np.array_equal(sample(int(np.rint(xu)), int(np.rint(xv))), got[py, px]):
        ok = [(du, dv) for du in (-1, 0, 1) for dv in (-1, 0, 1) if np.array_equal(sample(int(np.rint(xu)) + du, int(np.rint(xv)) + dv), got[py, px])]
        print("xu", repr(float(xu)), "xv", repr(float(xv)), "rint", int(np.rint(xu)), int(np.rint(xv)), "matching offsets", ok)
print("--- exact ties that rint got right")
for (u, v, gx, gy) in uvs:
    px, py = int((gx + 0.25) * W / G), int((gy + 0.25) * H / G)
    if not o[py, px].any(): continue
    xu = F(F(u * F(TW)) * F(256)); xv = F(F(v * F(TH)) * F(256))
    if (xu - np.floor(xu)) == 0.5 and np.array_equal(sample(int(np.rint(xu)), int(np.rint(xv))), got[py, px]):
        alt_ok = np.array_equal(sample(int(np.rint(xu)) + (1 if np.rint(xu) < xu else -1), int(np.rint(xv))), got[py, px])
        print("xu", float(xu), "-> ", int(np.rint(xu)), "(other neighbour also matches)" if alt_ok else "")
print("--- failing ties: alternative products")
for (u, v, gx, gy) in uvs:
    px, py = int((gx + 0.25) * W / G), int((gy + 0.25) * H / G)
    if not o[py, px].any(): continue
    xu = F(F(u * F(TW)) * F(256)); xv = F(F(v * F(TH)) * F(256))
    if not np.array_equal(sample(int(np.rint(xu)), int(np.rint(xv))), got[py, px]):
        print("u", repr(float(u)), "x2", float(xu), "u*1280 exact", float(np.float64(u) * 1280), "f32(u*1280)", float(F(u * F(1280))), "u*5 exact", float(np.float64(u) * 5), "f32", float(F(u * F(5))))
print("--- failing with the exact product")
for (u, v, gx, gy) in uvs:
    px, py = int((gx + 0.25) * W / G), int((gy + 0.25) * H / G)
    if not o[py, px].any(): continue
    eu, ev = np.float64(u) * (TW * 256), np.float64(v) * (TH * 256)
    if not np.array_equal(sample(int(np.rint(eu)), int(np.rint(ev))), got[py, px]):
        ok = [(du, dv) for du in (-1, 0, 1) for dv in (-1, 0, 1) if np.array_equal(sample(int(np.rint(eu)) + du, int(np.rint(ev)) + dv), got[py, px])]
        print("exact u*1280", float(eu), "exact v*1792", float(ev), "f32 v path", float(F(F(v * F(TH)) * F(256))), "matching offsets", ok)
