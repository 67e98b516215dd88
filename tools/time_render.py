#!/usr/bin/env python3
"""Time rt_render on device memory for a few scene/parameter variants (development aid)."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import api, scenes
from raytracer_amd.camera import Camera
from raytracer_amd.params import Uniforms

def empty_scene():
    z = np.zeros(3, np.float32)
    return dict(vertices=np.zeros((0, 4), np.float32), normals=np.zeros((0, 4), np.float32), tri_refs=np.zeros((0, 4), np.uint32), nodes=np.zeros(0, np.uint32), bounds_min=z, bounds_max=z)

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", default="empty,bunny")
    ap.add_argument("--limits", default="1,2,5")
    ap.add_argument("--width", type=int, default=3840); ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--count", action="store_true")
    ap.add_argument("--pipelines", default="wavefront,megakernel")
    ap.add_argument("--subframes", default="4", help="comma list of sub-frame counts")
    ap.add_argument("--order", default="reference", choices=["reference", "near_first", "near_first_exact", "auto"])
    ap.add_argument("--host", action="store_true", help="host-memory target: every frame is staged over PCIe")
    ap.add_argument("--sequence", action="store_true", help="issue the timed frames as one rt_render_sequence call")
    a = ap.parse_args()
    sky = scenes.procedural_skybox(900)
    W, H = a.width, a.height
    color = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    for name in a.scenes.split(","):
        d = empty_scene() if name == "empty" else {"bunny": scenes.default_octree, "s100k": scenes.synthetic_100k, "s1m": scenes.synthetic_1m, "small": scenes.small_test_scene}[name]().as_dict()
        sc = api.Scene.from_dict(d); sc.set_skybox(sky)
        for pipe, lim, sub in [(p, int(x), int(k)) for p in a.pipelines.split(",") for x in a.limits.split(",") for k in a.subframes.split(",")]:
            sc.set_pipeline(pipe); sc.set_subframes(sub); sc.set_traversal_order(a.order)
            host_color = np.zeros((H, W, 4), np.float32)
            def run(k, sync=False):
                if a.host:
                    u = Uniforms.from_camera(Camera(aspect=W / H), time=k / 60.0, max_recursion_limit=lim).pack()
                    sc.render(u, W, H, accum_index=k + 1, color=host_color, planes=False)
                    return
                u = Uniforms.from_camera(Camera(aspect=W / H), time=k / 60.0, max_recursion_limit=lim).pack()
                fr = api.make_frame(u, W, H, accum_index=k + 1, stream=torch.cuda.current_stream().cuda_stream)
                tg = api.rt_target(); tg.memory, tg.sync, tg.color = api.RT_MEM_DEVICE, int(bool(sync)), color.data_ptr()
                if sync is None: return fr, tg
                sc.render_raw(fr, tg)
            run(0); run(1); torch.cuda.synchronize()
            seq_frames = [run(k, None)[0] for k in range(a.reps)] if a.sequence and not a.host else None
            if seq_frames:
                sc.render_sequence_raw(seq_frames[:2], run(0, None)[1]); torch.cuda.synchronize()
            t0 = time.perf_counter()
            if seq_frames:
                sc.render_sequence_raw(seq_frames, run(0, None)[1])
            else:
                for k in range(a.reps): run(k)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / a.reps * 1e3
            extra = ""
            if a.count:
                sc.set_counting(True); run(0, True); st = sc.stats(); sc.set_counting(False)
                extra = f" rays={st['rays']} Mrays/s={st['rays'] / ms / 1e3:.0f} pops={st['node_pops']} tri={st['triangle_tests']} box={st['child_box_tests']} enter={st['rays_entering_octree']} fallbacks={st['exact_fallbacks']} mode={sc.last_mode()}"
            print(f"{name:8s} {pipe:10s} sub={sub} seq={int(a.sequence)} limit={lim} {W}x{H}: {ms:.3f} ms/frame{extra}", flush=True)
        sc.close()

if __name__ == "__main__":
    main()
