"""Probes of the mip-mapped lookup: the mip levels glGenerateMipmap leaves, and the level of detail llvmpipe computes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import refrun
F = np.float32
VERT = "#version 430\nlayout(location=0) in vec3 p; layout(location=1) in vec3 n; layout(location=2) in vec2 t; out vec3 vs_out_pos; out vec3 vs_out_norm; out vec2 vs_out_tex; uniform mat4 world; uniform mat4 worldIT; uniform mat4 viewProj;\nvoid main(){ gl_Position = vec4(p, 1); vs_out_pos = p; vs_out_norm = n; vs_out_tex = t; }\n"
def frag(body): return "#version 430\nin vec3 vs_out_pos; in vec3 vs_out_norm; in vec2 vs_out_tex; out vec4 c; layout(binding = 0) uniform sampler2D texImage;\nvoid main(){ " + body + " }\n"
I = np.eye(4, dtype=F)
words = refrun.overlay_uniform_words(I, [I], [I], [0, 0, -1], [0.25] * 3, [0.5] * 3)
rng = np.random.default_rng(3)
def full_screen(uv0, uv1):
    v = np.zeros((3, 8), F); v[:, :3] = [[-1, -1, 0], [3, -1, 0], [-1, 3, 0]]
    v[:, 6:8] = [[uv0[0], uv0[1]], [uv0[0] + 2 * (uv1[0] - uv0[0]), uv0[1]], [uv0[0], uv0[1] + 2 * (uv1[1] - uv0[1])]]
    return v
what = sys.argv[1] if len(sys.argv) > 1 else "levels"
if what == "levels":
    for (TH, TW) in ((8, 8), (6, 10), (7, 5)):
        tex = rng.integers(0, 256, (TH, TW, 4), dtype=np.uint8)
        W = H = 16
        lv = []
        for level in range(4):
            o = refrun.run_overlay(full_screen((0, 0), (1, 1)), np.arange(3, dtype=np.uint32), words, tex, W, H, vert_text=VERT,
                                   frag_text=frag("ivec2 s = textureSize(texImage, %d); ivec2 q = ivec2(gl_FragCoord.xy); c = (q.x < s.x && q.y < s.y) ? texelFetch(texImage, q, %d) : vec4(-1); if (q == ivec2(15, 15)) c = vec4(s, 0, 0);" % (level, level)))["rgba"]
            sw, sh = int(o[15, 15, 0]), int(o[15, 15, 1])
            lv.append(np.rint(o[:sh, :sw] * 255).astype(np.int64))
            print("texture", TW, "x", TH, "level", level, "size", sw, sh)
        # hypothesis: level L+1 = 8-bit bilinear lookup of level L at the centres of the smaller grid (clamp to edge)
        def lerp(a, b, w): return a + (((b - a) * w + 128) >> 8)
        for L in range(3):
            src, dst = lv[L], lv[L + 1]
            sh, sw = src.shape[:2]; dh, dw = dst.shape[:2]
            pred = np.zeros_like(dst)
            for j in range(dh):
                for i in range(dw):
                    su = int(np.rint(F(F((i + 0.5) / dw) * F(sw)) * F(256))) - 128; sv = int(np.rint(F(F((j + 0.5) / dh) * F(sh)) * F(256))) - 128
                    i0, wu, j0, wv = su >> 8, su & 255, sv >> 8, sv & 255
                    c = lambda a, n: min(max(a, 0), n - 1)
                    pred[j, i] = lerp(lerp(src[c(j0, sh), c(i0, sw)], src[c(j0, sh), c(i0 + 1, sw)], wu), lerp(src[c(j0 + 1, sh), c(i0, sw)], src[c(j0 + 1, sh), c(i0 + 1, sw)], wu), wv)
            box = None
            if sw == 2 * dw and sh == 2 * dh:
                s4 = src.reshape(dh, 2, dw, 2, 4)
                box = (s4.sum((1, 3)) + 2) >> 2
            print("   level", L + 1, "from", L, ": bilinear-8bit mismatches", int((pred != dst).sum()), "of", dst.size, "| box (sum+2)>>2 mismatches", None if box is None else int((box != dst).sum()))
if what == "lod":
    TW, TH = 64, 32
    tex = rng.integers(0, 256, (TH, TW, 4), dtype=np.uint8)
    W = H = 32
    for (du, dv) in ((2.0, 1.0), (3.0, 0.5), (1.5, 1.5), (5.0, 7.0), (1.0, 1.0), (0.3, 2.7), (1.37, 0.1), (10.0, 0.1)):
        # u advances du texels per pixel, v advances dv texels per pixel
        v = full_screen((0.0, 0.0), (du * W / TW, dv * H / TH))
        o = refrun.run_overlay(v, np.arange(3, dtype=np.uint32), words, tex, W, H, vert_text=VERT, frag_text=frag("c = vec4(textureQueryLod(texImage, vs_out_tex), vs_out_tex);"))["rgba"]
        lod = o[..., 1]; lvl = o[..., 0]
        uu, vv = o[..., 2], o[..., 3]
        ddx_u = (uu[:, 1::2] - uu[:, 0::2]) * TW; ddy_v = (vv[1::2, :] - vv[0::2, :]) * TH
        print(f"du {du} dv {dv}: lod min {lod.min():.6f} max {lod.max():.6f} distinct {len(np.unique(lod))}; level x min {lvl.min()} max {lvl.max()}; exact log2(max) {np.log2(max(du, dv)):.6f} log2(hypot) {np.log2(np.hypot(du, dv)):.6f}; measured ddx_u {ddx_u.min():.5f}..{ddx_u.max():.5f} ddy_v {ddy_v.min():.5f}..{ddy_v.max():.5f}")
def fast_log2(x):
    b = np.asarray(x, F).view(np.int32).astype(np.int64)
    return ((b - 0x3f800000).astype(np.float64) / 2 ** 23)
if what == "rho":
    TW, TH = 64, 32
    tex = rng.integers(0, 256, (TH, TW, 4), dtype=np.uint8)
    W = H = 32
    for M in (((2.0, 0.0), (0.0, 1.0)), ((2.0, 1.0), (0.5, 1.5)), ((1.0, 3.0), (2.0, 0.2)), ((0.7, 0.7), (-0.7, 0.7)), ((3.0, 0.0), (3.0, 0.0)), ((1.1, 2.3), (1.9, 0.4))):
        (a, b), (c, d) = M      # texels per pixel: du/dx = a, du/dy = b, dv/dx = c, dv/dy = d
        v = np.zeros((3, 8), F); v[:, :3] = [[-1, -1, 0], [3, -1, 0], [-1, 3, 0]]
        v[:, 6:8] = [[0, 0], [a * 2 * W / TW, c * 2 * W / TH], [b * 2 * H / TW, d * 2 * H / TH]]
        o = refrun.run_overlay(v, np.arange(3, dtype=np.uint32), words, tex, W, H, vert_text=VERT, frag_text=frag("c = vec4(textureQueryLod(texImage, vs_out_tex), vs_out_tex);"))["rgba"]
        lod = float(np.median(o[..., 1]))
        cands = {"max(|d/dx|^2, |d/dy|^2)": max(a * a + c * c, b * b + d * d), "max of the four squares": max(a * a, b * b, c * c, d * d), "max(max(|a|,|b|), max(|c|,|d|))^2": max(abs(a), abs(b), abs(c), abs(d)) ** 2,
                 "max(a^2+b^2, c^2+d^2)": max(a * a + b * b, c * c + d * d)}
        print(M, "lod", lod, {k: round(float(0.5 * fast_log2(F(r))), 6) for k, r in cands.items()})
if what == "lodbits":
    import itertools
    TW, TH = 64, 32
    tex = rng.integers(0, 256, (TH, TW, 4), dtype=np.uint8)
    W = H = 32
    v = np.zeros((3, 8), F); v[:, :3] = [[-1, -1, 0], [3, -1, 0], [-1, 3, 0]]
    v[:, 6:8] = [[0.013, 0.21], [4.37, 1.91], [-1.13, 5.77]]
    o = refrun.run_overlay(v, np.arange(3, dtype=np.uint32), words, tex, W, H, vert_text=VERT, frag_text=frag("c = vec4(textureQueryLod(texImage, vs_out_tex), vs_out_tex);"))["rgba"]
    lod, u, vv = o[..., 1], o[..., 2], o[..., 3]
    print("lod distinct per quad?", all(len(np.unique(lod[y:y+2, x:x+2])) == 1 for y in range(0, H, 2) for x in range(0, W, 2)))
    q = lambda a, dy, dx: a[dy::2, dx::2]
    u00, u10, u01 = q(u, 0, 0), q(u, 0, 1), q(u, 1, 0); v00, v10, v01 = q(vv, 0, 0), q(vv, 0, 1), q(vv, 1, 0)
    lq = q(lod, 0, 0)
    def flog2(x):
        b = np.asarray(x, F).view(np.int32)
        return (b - np.int32(0x3f800000)).astype(F) * F(2.0 ** -23)
    res = {}
    for scale_first, order in itertools.product((0, 1), (0, 1)):
        dudx, dudy, dvdx, dvdy = u10 - u00, u01 - u00, v10 - v00, v01 - v00
        if scale_first == 0:
            a, b, c, d = dudx * F(TW), dudy * F(TW), dvdx * F(TH), dvdy * F(TH)
            rx = a * a + c * c if order == 0 else c * c + a * a
            ry = b * b + d * d if order == 0 else d * d + b * b
        else:
            rx = (dudx * dudx) * F(TW * TW) + (dvdx * dvdx) * F(TH * TH); ry = (dudy * dudy) * F(TW * TW) + (dvdy * dvdy) * F(TH * TH)
        rho2 = np.maximum(rx, ry)
        for half in ("0.5*log", "log*0.5 after int"):
            l = flog2(rho2) * F(0.5)
            res[(scale_first, order, half)] = int((l.view(np.uint32) != lq.view(np.uint32)).sum())
    print(res, "of", lq.size)
    l = flog2(np.maximum((u10 - u00) * F(TW) * ((u10 - u00) * F(TW)) + (v10 - v00) * F(TH) * ((v10 - v00) * F(TH)), (u01 - u00) * F(TW) * ((u01 - u00) * F(TW)) + (v01 - v00) * F(TH) * ((v01 - v00) * F(TH)))) * F(0.5)
    print("sample", lq[3, :4], l[3, :4])
if what == "lodbits":
    u11, v11 = q(u, 1, 1), q(vv, 1, 1)
    def lod_of(dudx, dvdx, dudy, dvdy):
        a, b, c, d = dudx * F(TW), dudy * F(TW), dvdx * F(TH), dvdy * F(TH)
        return flog2(np.maximum(a * a + c * c, b * b + d * d)) * F(0.5)
    # per pixel: its own row for d/dx, its own column for d/dy
    pp = {(0, 0): lod_of(u10 - u00, v10 - v00, u01 - u00, v01 - v00), (0, 1): lod_of(u10 - u00, v10 - v00, u11 - u10, v11 - v10),
          (1, 0): lod_of(u11 - u01, v11 - v01, u01 - u00, v01 - v00), (1, 1): lod_of(u11 - u01, v11 - v01, u11 - u10, v11 - v10)}
    for (dy, dx), l in pp.items():
        print("pixel", dy, dx, "own row/column differences: mismatches", int((l.view(np.uint32) != q(lod, dy, dx).view(np.uint32)).sum()), "| top-left formula:", int((pp[(0, 0)].view(np.uint32) != q(lod, dy, dx).view(np.uint32)).sum()))
if what == "lodbits":
    bad = pp[(0, 0)].view(np.uint32) != q(lod, 1, 0).view(np.uint32)
    ys, xs = np.nonzero(bad)
    for y, x in zip(ys, xs):
        print("quad", x, y, "lods", lod[2*y:2*y+2, 2*x:2*x+2].ravel(), "formula", pp[(0,0)][y, x])
