#!/usr/bin/env python3
"""Headline benchmark of the MI355X ray-tracing pass.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[1]): the reference's default-style scene (bundled Stanford bunny standing in for the
missing dragon, material 1, scale 20, Octree(18,10,6,6); the shader's 84 spheres, cylinder and portals) at 3840x2160,
camera eye (2,2,2) -> origin, fovy 90 deg, "primary + shadow rays" mapped to max_recursion_limit = 2 (the reference
has no shadow rays: primary + one secondary segment), 1 sample per frame, frames progressively accumulated as
App::Render does (time advances by 1/60 s per frame).  One step = one frame.  Metric: Mrays/s where one ray is one
FindIntersection call (path segment); rays are COUNTED by replaying the timed frames with the library's work counters on.

N > 1: the frame is sharded by interleaved 16-row strips over the ranks (strong scaling), the scene is broadcast from
rank 0, every rank accumulates its own bands, and the accumulated frame is gathered to rank 0 with RCCL after every 16 x N
steps -- about 100 presentations per second whatever N is -- and after the last step, inside the timed region (on a side
stream while rendering goes on; the last gather is waited for before the clock stops).  A 4K RGBA32F frame is 132.7 MB:
gathered after every 0.1 ms step it would measure the xGMI links.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BAND_ROWS = 16
HBM_PEAK_GBS = 8000.0           # MI355X HBM3E peak
VALU_PEAK_TFLOPS_NO_FMA = 78.6  # 157.3 TFLOP/s fp32 vector peak counts an FMA as 2; this path may not fuse
CPU_BASELINE_SECONDS = 12.0     # the CPU leg renders whole frames of the same workload until this much time has passed


def traversal_bytes(st: dict) -> int:
    """Bytes the reference's data layout moves for the counted octree work: per node visit 2 header words + 4 B per
    child pointer; per triangle test a 16 B index record + three 16 B vertices."""
    return 8 * st["node_pops"] + 4 * st["child_box_tests"] + 64 * st["triangle_tests"]


def frame_bytes(st: dict, pixels: int, blended: bool) -> int:
    """traversal_bytes + per shaded triangle hit index + 3 vertices + 3 normals (112 B) + per pixel a 16 B colour
    write (+16 B read when blending with n > 1)."""
    return traversal_bytes(st) + 112 * st["triangle_hits_shaded"] + pixels * (16 + (16 if blended else 0))


def algorithmic_flops(st: dict) -> int:
    """Approximate arithmetic of the counted tests: sphere 17 flop x 84 per ray, child box 27, triangle 50."""
    return 84 * 17 * st["rays"] + 27 * st["child_box_tests"] + 50 * st["triangle_tests"]


def measured_traffic(kernel: str, default_workload: bool):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary of this command, or None (the summary is
    of the default single-GPU workload only)."""
    if not default_workload:
        return None
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(kernel, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--max-recursion-limit", type=int, default=2)
    ap.add_argument("--scene", default="bunny", choices=["bunny", "synthetic_100k", "synthetic_1m"])
    ap.add_argument("--pipeline", default="auto", choices=["auto", "persistent", "wavefront", "megakernel"])
    ap.add_argument("--subframes", type=int, default=None)
    ap.add_argument("--per-frame-calls", action="store_true", help="one rt_render per step instead of rt_render_sequence calls")
    ap.add_argument("--present-every", type=int, default=0,
                    help="N > 1: the accumulated frame is gathered to rank 0 after every this many steps and after the last one. "
                         "0 = 16 x N: a step takes about 0.67 ms / N, so the frame is presented about 100 times a second whatever N is "
                         "(a 4K RGBA32F frame is 132.7 MB: gathered after every step it would measure the xGMI links, not the renderer)")
    ap.add_argument("--traversal", default="reference", choices=["reference", "near_first"],
                    help="child order of the octree traversal: the shader's own (bit-identical results) or nearest first (faster; colour tolerance only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    # stdout carries exactly one JSON line: everything else that writes to file descriptor 1 (RCCL prints a version
    # banner there when it starts) goes to stderr for the rest of the run
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from raytracer_amd import api, scenes, dist as rdist
    from raytracer_amd.camera import Camera
    from raytracer_amd.params import Uniforms

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the ray-tracing pass has no CPU fallback")
    # RT_BENCH_REHEARSE=1: every rank on GPU 0 with the gloo backend -- a functional rehearsal of the N > 1 path on a
    # one-GPU box (RCCL does not accept two ranks on one device); never a measurement
    rehearse = os.environ.get("RT_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # RT_BENCH_FORCE_DIST=1 (with torch.distributed.run --nproc-per-node 1): the N > 1 code path -- RCCL process group,
    # scene broadcast, pipelined gather -- with a single rank, to exercise the RCCL calls on a one-GPU box
    use_dist = world > 1 or os.environ.get("RT_BENCH_FORCE_DIST") == "1"
    if use_dist:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    # ---- scene (built on rank 0 with the host-side builders, broadcast over RCCL) -------------------------------
    t_build = time.time()
    scene_arrays = None
    if rank == 0:
        oc = {"bunny": scenes.default_octree, "synthetic_100k": scenes.synthetic_100k, "synthetic_1m": scenes.synthetic_1m}[args.scene](device=local_rank)
        scene_arrays = oc.as_dict()
        scene_arrays["sky"] = scenes.procedural_skybox(900)
    if use_dist:
        scene_arrays = rdist.broadcast_scene(scene_arrays, device)
    sky = scene_arrays["sky"]
    sc = api.Scene.from_dict(scene_arrays, device=local_rank)
    sc.set_skybox(sky)
    sc.set_pipeline(args.pipeline)
    sc.set_traversal_order(args.traversal)
    if args.subframes:
        sc.set_subframes(args.subframes)
    t_build = time.time() - t_build

    W, H = args.width, args.height
    cam = Camera(aspect=W / H)
    def uniforms(step):
        return Uniforms.from_camera(cam, time=step / 60.0, max_recursion_limit=args.max_recursion_limit).pack()

    parts = world
    fr0 = api.make_frame(uniforms(0), W, H, band=(BAND_ROWS, parts, rank), compact=parts > 1)
    rows_local = api.band_row_count(fr0) if parts > 1 else H
    rows_pad = rdist.padded_row_count(H, BAND_ROWS, parts) if parts > 1 else H
    color = torch.zeros((rows_pad, W, 4), dtype=torch.float32, device=device)
    stream = torch.cuda.current_stream().cuda_stream

    def frame_of(step, accum_index):
        return api.make_frame(uniforms(step), W, H, accum_index=accum_index, blend=True, band=(BAND_ROWS, parts, rank), stream=stream, compact=parts > 1)

    def target_of(sync=False):
        tg = api.rt_target()
        tg.memory, tg.sync, tg.color = api.RT_MEM_DEVICE, int(sync), color.data_ptr()
        return tg

    def render(step, accum_index, sync=False):
        sc.render_raw(frame_of(step, accum_index), target_of(sync))

    gatherer = rdist.FrameGatherer(H, BAND_ROWS) if use_dist else None

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    # One step = one frame.  Frames are issued as the library's progressive-accumulation sequences (rt_render_sequence).
    # On several GPUs every rank accumulates its own bands; the accumulated frame is presented -- its bands gathered to
    # rank 0 on a side stream while rendering goes on -- after every `present_every` steps and after the last one.
    sequence = not args.per_frame_calls
    present_every = args.present_every if args.present_every > 0 else 16 * max(world, 1)

    def run_steps(frames):
        chunk = len(frames) if not use_dist else present_every
        for first in range(0, len(frames), max(chunk, 1)):
            part = frames[first:first + chunk]
            if sequence:
                sc.render_sequence_raw(part, target_of())
            else:
                for fr in part:
                    sc.render_raw(fr, target_of())
            if use_dist:
                gatherer.submit(color)
        if use_dist:
            gatherer.last()

    # the per-frame inputs (camera matrices, time) are prepared before the clock starts: they are the caller's data
    warm_frames = [frame_of(k, k + 1) for k in range(args.warmup)]
    timed_frames = [frame_of(k, k + 1) for k in range(args.steps)]

    # untimed: lets the library's pipeline selection settle on this workload (it times its own first frames)
    if args.pipeline == "auto":
        for k in range(12):
            if sequence:
                sc.render_sequence_raw([frame_of(4 * k + j, j + 1) for j in range(4)], target_of(sync=True))
            else:
                render(k, k + 1, sync=True)
    if args.warmup:
        run_steps(warm_frames)
    sync_all()
    pipeline, pipeline_ms = sc.pipeline_state()

    # the dominant kernel's launches are bracketed with HIP events on their own streams inside the library
    sc.set_kernel_timing(True)
    t0 = time.perf_counter()
    run_steps(timed_frames)
    sync_all()
    elapsed = time.perf_counter() - t0
    k_total_ms, k_launches = sc.kernel_time()
    sc.set_kernel_timing(False)
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- count the work of the timed frames (untimed replay with counters on) ----------------------------------
    sc.set_counting(True)
    tot = {}
    for k in range(args.steps):
        render(k, k + 1, sync=True)
        for name, v in sc.stats().items():
            tot[name] = tot.get(name, 0) + v
    sc.set_counting(False)
    local_tot = dict(tot)
    if use_dist:
        keys = sorted(tot)
        t = torch.tensor([tot[k] for k in keys], dtype=torch.int64, device=device)
        dist.all_reduce(t)
        tot = {k: int(v) for k, v in zip(keys, t.tolist())}

    rays = tot["rays"]
    value = rays / elapsed / 1e6
    out = {
        "metric": "Mrays/s (primary+secondary) & frame ms at 4K, 1/2/4/8 MI355X vs llvmpipe CPU",
        "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "strong" if world > 1 else "weak",
        "vs_baseline": None, "dtype": "f32", "data": ("REHEARSAL on one GPU, not a measurement; " if rehearse else "") + "synthetic camera/skybox; bundled bunny mesh (default scene's dragon is absent from the reference checkout)",
        "config": {"workload": f"{args.scene} default-style scene {W}x{H}, max_recursion_limit={args.max_recursion_limit} (primary + one secondary segment), 1 spp/frame, progressive accumulation",
                   "width": W, "height": H, "max_recursion_limit": args.max_recursion_limit, "band_rows": BAND_ROWS,
                   "parallelism": f"interleaved row bands x{world}" + (f" + RCCL gather of the accumulated frame every {present_every} steps (about 100 presentations per second) and after the last step" if world > 1 else "")},
        "rays_per_frame": rays / args.steps, "rays_per_pixel": rays / args.steps / (W * H),
        "work_per_frame": {k: v / args.steps for k, v in tot.items()},
        "scene_build_s": round(t_build, 3),
    }
    out["config"]["pipeline"] = pipeline + (" (selected by the library: measured ms/frame %s)" % {k: (round(v, 3) if v is not None else None) for k, v in pipeline_ms.items()} if args.pipeline == "auto" else "")
    out["config"]["traversal_order"] = args.traversal + (" (bit-identical to the reference shader)" if args.traversal == "reference" else " (not the reference's order: colour tolerance only)")
    out["config"]["issue"] = ("rt_render_sequence over the K steps" if world == 1 else f"rt_render_sequence per {present_every} steps") if sequence else "one rt_render per step"
    # ---- roofline of the dominant kernel, this rank's launches --------------------------------------------------
    kernel = {"wavefront": "traverse_kernel", "megakernel": "render_kernel", "persistent": "path_kernel"}[pipeline]
    if pipeline == "megakernel" and k_launches < args.steps:
        kernel = "render_batch_kernel"   # the library traced several frames of the sequence per launch
    k_ms = k_total_ms / max(k_launches, 1)
    default_workload = world == 1 and args.scene == "bunny" and (W, H, args.max_recursion_limit) == (3840, 2160, 2) and args.traversal == "reference"
    px_local = W * min(rows_local, H) if parts > 1 else W * H
    local_work = {k: v for k, v in local_tot.items()}
    total_bytes = traversal_bytes(local_work) if pipeline == "wavefront" else frame_bytes(local_work, px_local * args.steps, blended=True)
    abytes = total_bytes / max(k_launches, 1)
    achieved = abytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    out["roofline"] = {"kernel": kernel, "bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": measured_traffic(kernel, default_workload), "kernel_ms": round(k_ms, 4),
                       "launches": k_launches, "algorithmic_bytes_per_launch": int(abytes),
                       "launches_per_frame": k_launches / args.steps, "frames_per_launch": args.steps / max(k_launches, 1),
                       "device_rate": {"what": "algorithmic bytes of this kernel per frame / frame time (its launches of one frame run concurrently on separate streams)",
                                       "GB/s": round(total_bytes / max(k_launches, 1) * (k_launches / args.steps) / (1e6 * 1e3 * elapsed / args.steps), 3)},
                       "note": "octree traversal is bound by dependent-load latency and VALU issue, not by HBM bandwidth: see the valu object"}
    if achieved > HBM_PEAK_GBS:
        out["roofline"]["note"] += ("; frac > 1 here: the per-ray byte count prices every node and triangle record a ray touches as a memory read, and on this "
                                    "workload most of them are served by the caches, so the count is no lower bound on HBM bytes")
    aflops = algorithmic_flops(local_work) / args.steps
    frame_ms = 1e3 * elapsed / args.steps
    out["valu"] = {"scope": "whole frame" if world == 1 else "rank 0's share of the frame", "achieved_tflops": round(aflops / (frame_ms * 1e-3) / 1e12, 3), "peak_tflops_no_fma": VALU_PEAK_TFLOPS_NO_FMA,
                   "frac": round(aflops / (frame_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS_NO_FMA, 4), "algorithmic_flop_per_frame": int(aflops)}

    # ---- CPU baseline: the oracle (a port of the shader with the reference vehicle's numerics) on this host -----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle
        osc = oracle.OracleScene(scene_arrays, sky)
        threads = min(16, os.cpu_count() or 1)
        n_rays, n_frames = 0, 0
        t1 = time.perf_counter()
        while True:
            o = osc.render(uniforms(n_frames), W, H, taps=False, threads=threads)
            n_rays += int(o["counters"]["rays"])
            n_frames += 1
            dt = time.perf_counter() - t1
            if dt >= CPU_BASELINE_SECONDS:
                break
        out["cpu_baseline"] = {"value": round(n_rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
                               "sample": f"{n_frames} consecutive {W}x{H} frames of the same workload (time = k/60 s; {n_rays} rays) in {dt:.1f} s"}
    if rank == 0:
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
