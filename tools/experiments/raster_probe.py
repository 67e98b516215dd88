"""Development probe: random triangles through llvmpipe (reference raster shaders or probe shaders) against a numpy model of the
rasteriser, to find the arithmetic that reproduces coverage and the 24-bit depth.  python tools/experiments/raster_probe.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import refrun

F = np.float32
def model(clip, W, H, variant):
    """clip: [3,4] float32 clip-space vertices of one triangle. Returns (mask[H,W], depth24[H,W])."""
    w = clip[:, 3]
    if variant.get("rcp_w", True):
        oow = F(1.0) / w
        ndc = clip[:, :3] * oow[:, None]
    else:
        ndc = clip[:, :3] / w[:, None]
    sx, sy = F(W / 2), F(H / 2)
    xw = ndc[:, 0] * sx + sx; yw = ndc[:, 1] * sy + sy; zw = ndc[:, 2] * F(0.5) + F(0.5)
    fx = np.rint((xw - F(0.5)) * F(256)).astype(np.int64); fy = np.rint((yw - F(0.5)) * F(256)).astype(np.int64)
    area = (fx[0] - fx[1]) * (fy[2] - fy[0]) - (fx[0] - fx[2]) * (fy[1] - fy[0])   # sign convention found by probing
    area2 = (fx[1] - fx[0]) * (fy[2] - fy[0]) - (fx[2] - fx[0]) * (fy[1] - fy[0])
    mask = np.zeros((H, W), bool); d24 = np.zeros((H, W), np.uint32)
    if area2 <= 0:      # GL: counter-clockwise is front; back faces culled
        return mask, d24
    px = np.arange(W, dtype=np.int64) * 256; py = np.arange(H, dtype=np.int64) * 256
    X, Y = np.meshgrid(px, py)
    inside = np.ones((H, W), bool)
    for i in range(3):
        j = (i + 1) % 3
        ex, ey = fx[j] - fx[i], fy[j] - fy[i]
        e = ex * (Y - fy[i]) - ey * (X - fx[i])          # > 0 inside for ccw
        rule = variant.get("rule", "topleft")
        # tie on the edge: included for "top" and "left" edges
        if rule == "topleft":
            incl = (ey < 0) or (ey == 0 and ex > 0)      # placeholder, fixed by probing
        elif rule == "alt":
            incl = (ey > 0) or (ey == 0 and ex < 0)
        elif rule == "alt2":
            incl = (ey < 0) or (ey == 0 and ex < 0)
        else:
            incl = (ey > 0) or (ey == 0 and ex > 0)
        inside &= (e > 0) | ((e == 0) & incl)
    mask = inside
    # depth plane from float window coordinates
    x0, y0, z0 = xw[0], yw[0], zw[0]
    dx01, dy01, dz01 = xw[0] - xw[1], yw[0] - yw[1], zw[0] - zw[1]
    dx20, dy20, dz20 = xw[2] - xw[0], yw[2] - yw[0], zw[2] - zw[0]
    ooa = F(1.0) / (dx01 * dy20 - dx20 * dy01)
    dzdx = (dz01 * (dy20 * ooa)) - ((dy01 * ooa) * dz20)
    dzdy = (dz20 * (dx01 * ooa)) - ((dx20 * ooa) * dz01)
    a0 = z0 - (dzdx * (x0 - F(0.5)) + dzdy * (y0 - F(0.5)))
    Xf = np.arange(W, dtype=F)[None, :]; Yf = np.arange(H, dtype=F)[:, None]
    zi = variant.get("z", "plain")
    def fma(a, b, c): return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)
    if zi == "plain": z = (a0 + dzdx * Xf) + dzdy * Yf
    elif zi == "fma_xy": z = fma(np.broadcast_to(dzdy, (H, W)), np.broadcast_to(Yf, (H, W)), fma(np.broadcast_to(dzdx, (H, W)), np.broadcast_to(Xf, (H, W)), np.broadcast_to(a0, (H, W))))
    elif zi == "fma_yx": z = fma(np.broadcast_to(dzdx, (H, W)), np.broadcast_to(Xf, (H, W)), fma(np.broadcast_to(dzdy, (H, W)), np.broadcast_to(Yf, (H, W)), np.broadcast_to(a0, (H, W))))
    elif zi == "sum_first": z = a0 + (dzdx * Xf + dzdy * Yf)
    elif zi == "fma_sum": z = a0 + fma(np.broadcast_to(dzdy, (H, W)), np.broadcast_to(Yf, (H, W)), np.broadcast_to(dzdx * Xf, (H, W)))
    z = np.clip(z, F(0), F(1))
    d24 = np.rint((z * F(16777215.0)).astype(F)).astype(np.uint32)
    return mask, d24

def run(n=40, W=64, H=64, seed=0, persp=False):
    rng = np.random.default_rng(seed)
    I = np.eye(4, dtype=F)
    words = refrun.overlay_uniform_words(I, [I], [I], [0, 0, -1], [0.25] * 3, [0.5] * 3)
    tex = np.full((4, 4, 4), 255, np.uint8)
    tot = {}
    for k in range(n):
        clip = np.ones((3, 4), F)
        clip[:, :3] = rng.uniform(-1, 1, (3, 3)).astype(F)
        if persp:
            clip[:, 3] = rng.uniform(0.5, 3.0, 3).astype(F); clip[:, :3] *= clip[:, 3:4]
        if k % 5 == 0:      # vertices on pixel centres / corners: edges pass exactly through sample points
            clip[:, :2] = (rng.integers(0, W * 2 + 1, (3, 2)) / F(W) - 1).astype(F)
        v = np.zeros((3, 8), F); v[:, :3] = clip[:, :3]; v[:, 3:6] = [0, 0, 1]
        if persp:
            # w through a probe vertex shader: position.xyz = clip.xyz, texcoord.x = w
            v[:, 6] = clip[:, 3]
            vert = "#version 430\nlayout(location=0) in vec3 p; layout(location=1) in vec3 n; layout(location=2) in vec2 t; out vec3 vs_out_pos; out vec3 vs_out_norm; out vec2 vs_out_tex; uniform mat4 world; uniform mat4 worldIT; uniform mat4 viewProj;\nvoid main(){ gl_Position = vec4(p, t.x); vs_out_pos = p; vs_out_norm = n; vs_out_tex = vec2(0); }\n"
        else:
            vert = None
        o = refrun.run_overlay(v, np.array([0, 1, 2], np.uint32), words, tex, W, H, vert_text=vert)
        ref_mask = o["rgba"][..., 3] > 0
        ref_d24 = np.rint(o["depth"].astype(np.float64) * 16777215.0).astype(np.uint32)
        for name, var in VARIANTS.items():
            m, d = model(clip, W, H, var)
            cov = int((m != ref_mask).sum())
            both = m & ref_mask
            dd = int((d[both] != ref_d24[both]).sum())
            t = tot.setdefault(name, [0, 0, 0, 0])
            t[0] += cov; t[1] += dd; t[2] += int(both.sum()); t[3] += int(ref_mask.sum())
    for name, t in tot.items():
        print(f"{name:12s} coverage mismatches {t[0]:6d}   depth mismatches {t[1]:6d} of {t[2]} (covered by llvmpipe: {t[3]})")

VARIANTS = {z: dict(rule="topleft", z=z) for z in ("plain", "fma_xy", "fma_yx", "sum_first", "fma_sum")}
if __name__ == "__main__":
    run(persp="persp" in sys.argv)
