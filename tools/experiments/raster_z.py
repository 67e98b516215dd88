import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import refrun
F = np.float32
FRAG = "#version 430\nin vec3 vs_out_pos; in vec3 vs_out_norm; in vec2 vs_out_tex; out vec4 c;\nvoid main(){ c = vec4(gl_FragCoord.z, gl_FragCoord.w, gl_FragCoord.x, 1.0); }\n"
rng = np.random.default_rng(3)
I = np.eye(4, dtype=F)
words = refrun.overlay_uniform_words(I, [I], [I], [0, 0, -1], [0.25] * 3, [0.5] * 3)
tex = np.full((4, 4, 4), 255, np.uint8)
W = H = 32
clip = np.ones((3, 4), F)
clip[:, :3] = np.array([[-0.83, -0.71, 0.31], [0.77, -0.52, -0.4], [0.12, 0.9, 0.66]], F)
v = np.zeros((3, 8), F); v[:, :3] = clip[:, :3]
o = refrun.run_overlay(v, np.array([0, 1, 2], np.uint32), words, tex, W, H, frag_text=FRAG)
m = o["rgba"][..., 3] > 0
z = o["rgba"][..., 0]
ys, xs = np.nonzero(m)
print(m.sum())
# fit: is z affine in x,y exactly in float? look at finite differences along a row
row = ys[len(ys) // 2]
cols = xs[ys == row]
print("row", row, "cols", cols[:8])
zz = z[row, cols]
print([hex(v) for v in zz[:6].view(np.uint32)])
print(np.diff(zz.astype(np.float64))[:6])
sx = F(W / 2)
xw = clip[:, 0] * sx + sx; yw = clip[:, 1] * sx + sx; zw = clip[:, 2] * F(0.5) + F(0.5)
print("window", xw, yw, zw)
np.savez("/tmp/rz.npz", z=z, m=m, clip=clip)
