import itertools, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import refrun
F = np.float32
VERT = "#version 430\nlayout(location=0) in vec3 p; layout(location=1) in vec3 n; layout(location=2) in vec2 t; out vec3 vs_out_pos; out vec3 vs_out_norm; out vec2 vs_out_tex; uniform mat4 world; uniform mat4 worldIT; uniform mat4 viewProj;\nvoid main(){ gl_Position = vec4(p, 1); vs_out_pos = p; vs_out_norm = n; vs_out_tex = t; }\n"
FRAG_T = "#version 430\nin vec3 vs_out_pos; in vec3 vs_out_norm; in vec2 vs_out_tex; out vec4 c; layout(binding = 0) uniform sampler2D texImage;\nvoid main(){ c = texture(texImage, vs_out_tex); }\n"
FRAG_UV = "#version 430\nin vec3 vs_out_pos; in vec3 vs_out_norm; in vec2 vs_out_tex; out vec4 c; layout(binding = 0) uniform sampler2D texImage;\nvoid main(){ c = vec4(vs_out_tex, textureQueryLod(texImage, vs_out_tex).y, 1); }\n"
rng = np.random.default_rng(1)
TW, TH = 8, 4
tex = rng.integers(0, 256, (TH, TW, 4), dtype=np.uint8)
I = np.eye(4, dtype=F)
words = refrun.overlay_uniform_words(I, [I], [I], [0, 0, -1], [0.25] * 3, [0.5] * 3)
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3     # uv range over the 64-pixel viewport: 0.3 * 8 texels over 64 px = magnification
v = np.zeros((3, 8), F); v[:, :3] = [[-1, -1, 0], [3, -1, 0], [-1, 3, 0]]; v[:, 6:8] = np.array([[0.1, 0.2], [0.1 + 2 * scale, 0.2], [0.1, 0.2 + 2 * scale]], F) - 0.3
W = H = 64
ot = refrun.run_overlay(v, np.array([0, 1, 2], np.uint32), words, tex, W, H, frag_text=FRAG_T, vert_text=VERT)["rgba"]
ouv = refrun.run_overlay(v, np.array([0, 1, 2], np.uint32), words, tex, W, H, frag_text=FRAG_UV, vert_text=VERT)["rgba"]
print("lod range", ouv[..., 2].min(), ouv[..., 2].max())
u, vv = ouv[..., 0], ouv[..., 1]
got = np.rint(ot * 255).astype(np.int32)
print("max deviation of llvmpipe's output from k/255:", np.abs(ot * 255 - got).max())
t = tex.astype(np.int64)
def sample(mode):
    fx_mode, lerp_mode = mode
    if fx_mode == 0:      # fixed point: floor(u * size * 256) - 128
        su = np.floor(u.astype(F) * F(TW) * F(256)).astype(np.int64) - 128; sv = np.floor(vv.astype(F) * F(TH) * F(256)).astype(np.int64) - 128
    elif fx_mode == 1:    # round
        su = np.rint(u.astype(F) * F(TW) * F(256)).astype(np.int64) - 128; sv = np.rint(vv.astype(F) * F(TH) * F(256)).astype(np.int64) - 128
    else:                 # float: floor(u*size - 0.5), weight = floor(frac * 256)
        a = u.astype(F) * F(TW) - F(0.5); b = vv.astype(F) * F(TH) - F(0.5)
        su = np.floor(a).astype(np.int64) * 256 + np.floor((a - np.floor(a)) * F(256)).astype(np.int64); sv = np.floor(b).astype(np.int64) * 256 + np.floor((b - np.floor(b)) * F(256)).astype(np.int64)
    i0, wu = su >> 8, su & 255; j0, wv = sv >> 8, sv & 255
    i1, j1 = i0 + 1, j0 + 1
    i0 %= TW; i1 %= TW; j0 %= TH; j1 %= TH
    def lerp(a, b, w):
        w = w[..., None]
        if lerp_mode == 0: return a + (((b - a) * w) >> 8)
        if lerp_mode == 1: return a + (((b - a) * w + 128) >> 8)
        if lerp_mode == 2:
            w2 = w + (w >> 7); return a + (((b - a) * w2) >> 8)
        if lerp_mode == 3:
            w2 = w + (w >> 7); return a + (((b - a) * w2 + 128) >> 8)
        if lerp_mode == 4:
            return (a * (256 - w) + b * w + 128) >> 8
        if lerp_mode == 5:
            return (a * (256 - w) + b * w) >> 8
    top = lerp(t[j0, i0], t[j0, i1], wu); bot = lerp(t[j1, i0], t[j1, i1], wu)
    return lerp(top, bot, wv)
for mode in itertools.product(range(3), range(6)):
    s = sample(mode)
    print(mode, int((s != got).sum()), int(np.abs(s - got).max()))
