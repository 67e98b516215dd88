"""Development check: oracle/raster_oracle.c against llvmpipe running the reference's raster shaders."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import refrun, raster
F = np.float32
def compare(v, idx, words, tex, W, H, label):
    o = refrun.run_overlay(v, idx, words, tex, W, H)
    col = np.zeros((H, W, 4), F); d24 = np.full((H, W), 0xFFFFFF, np.uint32)
    raster.draw(v, idx, words, tex, col, d24)
    ref24 = np.rint(o["depth"].astype(np.float64) * 16777215.0).astype(np.uint32)
    cov_ref = ref24 != 0xFFFFFF; cov = d24 != 0xFFFFFF
    nd = int((d24 != ref24).sum()); nc = int((col.view(np.uint32) != o["rgba"].view(np.uint32)).any(-1).sum())
    err = np.abs(col.astype(np.float64) - o["rgba"]).max()
    print(f"{label}: covered {int(cov_ref.sum())}  coverage diff {int((cov != cov_ref).sum())}  depth24 diff {nd}  colour pixels differing {nc} (max abs {err:.3g})")
    return o, col, d24
if __name__ == "__main__":
    rng = np.random.default_rng(0)
    tex = rng.integers(0, 256, (16, 16, 4), dtype=np.uint8); tex[..., 3] = 255
    I = np.eye(4, dtype=F)
    # random soups, orthographic and perspective, with texture coordinates in the magnification range
    for k in range(6):
        n = 40
        v = np.zeros((3 * n, 8), F)
        c = rng.uniform(-0.6, 0.6, (n, 1, 3)); v[:, :3] = (c + rng.uniform(-0.39, 0.39, (n, 3, 3))).reshape(-1, 3)
        v[:, 3:6] = rng.uniform(-1, 1, (3 * n, 3)); v[:, 6:8] = rng.uniform(0, 0.05, (3 * n, 2)) + rng.uniform(-2, 2, (n, 1, 2)).repeat(3, 1).reshape(-1, 2)
        if k % 2:
            from raytracer_amd.camera import Camera
            cam = Camera(eye=(0.3, 0.2, 2.5), at=(0, 0, 0), fovy=1.0, aspect=1.5, z_near=0.5, z_far=20.0)
            VP = np.asarray(cam.view_proj(), F).T.reshape(-1) if hasattr(cam, "view_proj") else None
        else:
            VP = I.reshape(-1)
        words = raster.overlay_words(VP, [I.reshape(-1)], [I.reshape(-1)], (-1, -2, -0.5), (0.1, 0.2, 0.3), (0.9, 0.8, 0.7))
        compare(v, np.arange(3 * n, dtype=np.uint32), words, tex, 96, 64, f"soup {k}")
    print("--- clipping stress, constant texcoord")
    from raytracer_amd.camera import Camera
    for k in range(8):
        n = 30
        v = np.zeros((3 * n, 8), F)
        c = rng.uniform(-1.5, 1.5, (n, 1, 3)); v[:, :3] = (c + rng.uniform(-1.2, 1.2, (n, 3, 3))).reshape(-1, 3)
        v[:, 3:6] = rng.uniform(-1, 1, (3 * n, 3))
        if k % 2:
            cam = Camera(eye=(0.3, 0.2, 1.0), at=(0, 0, 0), fovy=1.2, aspect=1.5, z_near=0.3, z_far=3.0)
            VP = np.asarray(cam.view_proj(), F).T.reshape(-1)
        else:
            VP = I.reshape(-1)
        words = raster.overlay_words(VP, [I.reshape(-1)], [I.reshape(-1)], (-1, -2, -0.5), (0.1, 0.2, 0.3), (0.9, 0.8, 0.7))
        compare(v, np.arange(3 * n, dtype=np.uint32), words, tex, 96, 64, f"clip soup {k}")
