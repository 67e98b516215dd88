import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import refrun
F = np.float32
HEAD = "#version 430\nin vec3 vs_out_pos; in vec3 vs_out_norm; in vec2 vs_out_tex; out vec4 c; uniform vec3 light_dir; uniform vec3 La; uniform vec3 Ld; layout(binding = 0) uniform sampler2D texImage;\n"
F_IN = HEAD + "void main(){ c = vec4(vs_out_norm, 1); }\n"
F_N = HEAD + "void main(){ c = vec4(normalize(vs_out_norm), 1); }\n"
F_L = HEAD + "void main(){ c = vec4(normalize(-light_dir), 1); }\n"
F_C = HEAD + "void main(){ vec3 normal = normalize( vs_out_norm ); vec3 to_light = normalize( -light_dir ); float cosa = clamp(dot(normal, to_light), 0, 1); c = vec4(cosa, dot(normal, to_light), 0, 1); }\n"
F_D = HEAD + "void main(){ vec3 normal = normalize( vs_out_norm ); vec3 to_light = normalize( -light_dir ); float cosa = clamp(dot(normal, to_light), 0, 1); vec3 diffuse = cosa*Ld; c = vec4(La + diffuse, 1); }\n"
F_T = HEAD + "void main(){ c = texture(texImage, vs_out_tex); }\n"
rng = np.random.default_rng(4)
I = np.eye(4, dtype=F)
light, la, ld = np.array([-1, -2, -0.5], F), np.array([0.1, 0.2, 0.3], F), np.array([0.9, 0.8, 0.7], F)
words = refrun.overlay_uniform_words(I, [I], [I], light, la, ld)
tex = rng.integers(0, 256, (16, 16, 4), dtype=np.uint8)
v = np.zeros((3, 8), F); v[:, :3] = [[-1, -1, 0.3], [1, -0.8, -0.2], [0.1, 1, 0.5]]; v[:, 3:6] = rng.uniform(-1, 1, (3, 3)); v[:, 6:8] = rng.uniform(0, 0.05, (3, 2))
def run(fr): return refrun.run_overlay(v, np.array([0, 1, 2], np.uint32), words, tex, 48, 48, frag_text=fr)["rgba"]
vin = run(F_IN); m = vin[..., 3] > 0
n_in = vin[m][:, :3]
def bits(a): return np.ascontiguousarray(a, F).view(np.uint32)
def cmp(name, mine, ref): print(f"{name:40s} mismatching values {int((bits(mine) != bits(ref)).sum())} of {mine.size}")
dot = lambda a, b: (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]
gn = run(F_N)[m][:, :3]
d = dot(n_in, n_in)
cmp("normalize: v * (1/sqrt(d))", n_in * (F(1) / np.sqrt(d))[:, None], gn)
cmp("normalize: v / sqrt(d)", n_in / np.sqrt(d)[:, None], gn)
def rsq_nr(x):      # rsqrtps is not reproducible in numpy; placeholder
    return None
gl = run(F_L)[m][0, :3]
ml = -light; l1 = ml * (F(1) / np.sqrt(dot(ml, ml))); l2 = ml / np.sqrt(dot(ml, ml))
print("to_light", gl, bits(gl), bits(l1), bits(l2))
gc = run(F_C)[m]
nn = gn; dd = dot(nn, np.broadcast_to(gl, nn.shape))
cmp("dot(normal, to_light)", dd, gc[:, 1])
dd2 = (nn[:, 2] * gl[2] + nn[:, 1] * gl[1]) + nn[:, 0] * gl[0]
cmp("dot reversed order", dd2, gc[:, 1])
cmp("clamp", np.minimum(np.maximum(gc[:, 1], F(0)), F(1)), gc[:, 0])
gd = run(F_D)[m][:, :3]
cmp("La + cosa*Ld", la + gc[:, :1] * ld, gd)
def fma(a, b, c): return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(F)
cmp("fma(cosa, Ld, La)", fma(gc[:, :1], ld, la), gd)
d2 = (n_in[:, 2] * n_in[:, 2] + n_in[:, 1] * n_in[:, 1]) + n_in[:, 0] * n_in[:, 0]
cmp("normalize rev: v * (1/sqrt(d2))", n_in * (F(1) / np.sqrt(d2))[:, None], gn)
cmp("normalize rev: v / sqrt(d2)", n_in / np.sqrt(d2)[:, None], gn)
d3 = (n_in[:, 1] * n_in[:, 1] + n_in[:, 2] * n_in[:, 2]) + n_in[:, 0] * n_in[:, 0]
cmp("normalize yzx", n_in * (F(1) / np.sqrt(d3))[:, None], gn)
d4 = (n_in[:, 0] * n_in[:, 0] + n_in[:, 2] * n_in[:, 2]) + n_in[:, 1] * n_in[:, 1]
cmp("normalize xzy", n_in * (F(1) / np.sqrt(d4))[:, None], gn)
cmp("normalize fma chain", n_in * (F(1) / np.sqrt(fma(n_in[:, 2], n_in[:, 2], fma(n_in[:, 1], n_in[:, 1], n_in[:, 0] * n_in[:, 0]))))[:, None], gn)
gt = run(F_T)[m]
k = np.rint(gt * 255).astype(np.int32)
cmp("tex = k / 255", (k.astype(F) / F(255)), gt)
cmp("tex = k * (1/255)", (k.astype(F) * (F(1) / F(255))), gt)
cmp("tex = k * 0.00392156886", (k.astype(F) * F(0.00392156886)), gt)
F_ALL = HEAD + "void main(){ vec3 normal = normalize( vs_out_norm ); vec3 to_light = normalize( -light_dir ); float cosa = clamp(dot(normal, to_light), 0, 1); vec3 diffuse = cosa*Ld; c = vec4(La + diffuse, 1) * texture(texImage, vs_out_tex); }\n"
ga = run(F_ALL)[m]
cmp("(La + diffuse) * tex", gd * gt[:, :3], ga[:, :3])
cmp("La*tex + diffuse*tex ", la * gt[:, :3] + (gc[:, :1] * ld) * gt[:, :3], ga[:, :3])
cmp("fma(cosa*Ld... ", fma(gc[:, :1] * ld, gt[:, :3], la * gt[:, :3]), ga[:, :3])
