"""Times rt_overlay_draw (the reference's nine draws of its mesh, App.cpp:147-164) on device-resident planes:
python tools/time_overlay.py [--width 3840 --height 2160 --calls 50 --view front|close]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=3840); ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--calls", type=int, default=50); ap.add_argument("--view", default="front", choices=["front", "close"])
    a = ap.parse_args()
    import torch
    from raytracer_amd import api
    from raytracer_amd.camera import Camera
    from oracle import raster      # (only the instance matrices and the uniform layout: nothing is computed by the oracle here)
    z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "overlay_assets.npz"))
    w, h = a.width, a.height
    cam = Camera(eye=(-2.5, 5.25, 11.0), at=(-2.5, 5.25, 4.1), aspect=w / h) if a.view == "front" else Camera(eye=(-1.9, 5.55, 5.25), at=(-2.5, 5.25, 4.1), aspect=w / h)
    worlds, its = raster.reference_instances()
    m = api.OverlayMesh(z["vertices"], z["indices"], z["texture"])
    col = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda"); d24 = torch.full((h, w), 0xFFFFFF, dtype=torch.int32, device="cuda")
    args = (cam.view_proj().ravel(), worlds, its, raster.REFERENCE_LIGHT["light_dir"], raster.REFERENCE_LIGHT["la"], raster.REFERENCE_LIGHT["ld"])
    for _ in range(3):
        d24.fill_(0xFFFFFF); m.draw(col, d24, *args)
    torch.cuda.synchronize()
    covered = int((d24 != 0xFFFFFF).sum())
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(a.calls):
        m.draw(col, d24, *args)
    t1.record(); torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / a.calls
    n_tri = 9 * len(z["indices"]) // 3
    # bytes the pass has to move at least: the depth plane read and the key plane written and read (4 + 8 + 8 per pixel), colour and depth written per covered pixel
    algo = w * h * 20 + covered * 20
    print(json.dumps({"what": "rt_overlay_draw", "view": a.view, "width": w, "height": h, "triangles": n_tri, "covered_pixels": covered, "ms_per_call": round(ms, 4),
                      "algorithmic_GB": round(algo / 1e9, 4), "GBps": round(algo / 1e9 / (ms / 1e3), 1)}))
    m.close()


if __name__ == "__main__":
    main()
