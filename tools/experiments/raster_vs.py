import itertools, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import refrun
F = np.float32
which = sys.argv[1] if len(sys.argv) > 1 else "pos"
expr = {"pos": "viewProj * world * vec4( p, 1 )", "wpos": "vec4((world * vec4(p, 1)).xyz, 1)", "norm": "vec4((worldIT * vec4(n, 0)).xyz, 1)"}[which]
VERT = "#version 430\nlayout(location=0) in vec3 p; layout(location=1) in vec3 n; layout(location=2) in vec2 t; out vec3 vs_out_pos; out vec3 vs_out_norm; out vec2 vs_out_tex; flat out vec4 tap; uniform mat4 world; uniform mat4 worldIT; uniform mat4 viewProj;\nvoid main(){ gl_Position = vec4(t.x * 0.9, t.y * 0.9, 0, 1); vs_out_pos = p; vs_out_norm = n; vs_out_tex = vec2(0); tap = %s; }\n" % expr
FRAG = "#version 430\nin vec3 vs_out_pos; in vec3 vs_out_norm; in vec2 vs_out_tex; flat in vec4 tap; out vec4 c;\nvoid main(){ c = tap; }\n"
rng = np.random.default_rng(0)
tex = np.full((4, 4, 4), 255, np.uint8)
res = []
for k in range(12):
    VP = rng.uniform(-2, 2, (4, 4)).astype(F)      # VP[c] = column c
    if k % 2 == 0:
        Wm = np.eye(4, dtype=F); Wm[3, :3] = rng.uniform(-8, 8, 3).astype(F)      # column-major: column 3 = translation
    else:
        Wm = rng.uniform(-2, 2, (4, 4)).astype(F)
    WI = rng.uniform(-2, 2, (4, 4)).astype(F)
    words = refrun.overlay_uniform_words(VP.reshape(-1), [Wm.reshape(-1)], [WI.reshape(-1)], [0, 0, -1], [0.25] * 3, [0.5] * 3)
    p = rng.uniform(-1.5, 1.5, 3).astype(F); n = rng.uniform(-1, 1, 3).astype(F)
    v = np.zeros((3, 8), F); v[:, :3] = p; v[:, 3:6] = n; v[:, 6:8] = [[-1, -1], [1, -1], [0, 1]]
    o = refrun.run_overlay(v, np.array([0, 1, 2], np.uint32), words, tex, 8, 8, frag_text=FRAG, vert_text=VERT)
    got = o["rgba"][3, 4]
    res.append((VP, Wm, WI, p, n, got))
def mv(M, v, order):      # M[c] column c; order of accumulation
    terms = [M[c] * v[c] for c in range(4)]
    acc = terms[order[0]]
    for c in order[1:]: acc = acc + terms[c]
    return acc
def mm(A, B, order): return np.stack([mv(A, B[c], order) for c in range(4)])
best = {}
for o1 in itertools.permutations(range(4)):
    for o2 in itertools.permutations(range(4)):
        for assoc in ("(VP*W)*v", "VP*(W*v)"):
            bad = 0
            for VP, Wm, WI, p, n, got in res:
                if which == "pos":
                    v4 = np.array([p[0], p[1], p[2], 1], F)
                    r = mv(mm(VP, Wm, o1), v4, o2) if assoc == "(VP*W)*v" else mv(VP, mv(Wm, v4, o1), o2)
                elif which == "wpos":
                    r = mv(Wm, np.array([p[0], p[1], p[2], 1], F), o1); r = np.array([r[0], r[1], r[2], 1], F)
                else:
                    r = mv(WI, np.array([n[0], n[1], n[2], 0], F), o1); r = np.array([r[0], r[1], r[2], 1], F)
                bad += int((r.view(np.uint32) != got.view(np.uint32)).sum())
            best[(o1, o2, assoc)] = bad
for k, b in sorted(best.items(), key=lambda kv: kv[1])[:6]: print(k, b)
