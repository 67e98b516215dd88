#!/usr/bin/env python3
"""Headline benchmark of the MI355X ray-tracing pass.

  python bench.py [--gpus N] [--steps K] [--warmup W]

With --gpus N > 1 and no launcher in the environment the script starts its own N ranks (a child
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py ...`, before anything
in this process touches a GPU) and relays rank 0's JSON line; started under torch.distributed.run it is one of the ranks.

Workload (BASELINE.json configs[1]): the reference's default scene with the bundled Stanford bunny standing in for the
missing dragon (material 1, scale 20, Octree(18,10,6,6); the shader's 84 spheres, cylinder and portals) at 3840x2160,
camera eye (2,2,2) -> origin, fovy 90 deg, "primary + shadow rays" mapped to max_recursion_limit = 2 (the reference
has no shadow rays: primary + one secondary segment), 1 sample per frame, frames progressively accumulated as
App::Render does (time advances by 1/60 s per frame).  One step = one frame.  Metric: Mrays/s where one ray is one
FindIntersection call (path segment); rays are COUNTED by replaying the timed frames with the library's work counters on.

The same JSON line carries short sub-reports of the two deep scenes, measured the same way right after the headline workload
on the same ranks, each with the library's default settings (pipeline and child order selected by the library; outputs
bit-identical to the reference shader's) and with the shader's own child order forced:
  north_star / north_star_reference_order   BASELINE.json configs[3]: the seeded 1.28 M-triangle scene (depth-18 octree), limit 5
  config3 / config3_reference_order         BASELINE.json configs[2]: the seeded 100 000-primitive scene (depth-8 octree), limit 5
`frame_crc32` is the CRC of the accumulated frame after the K steps: equal between the two orders of a scene.

The line is kept short (the driver stores a tail of stdout); everything else -- per-frame work counters, the counter digest,
the load-path figures -- goes to stderr as one line starting with "BENCH_DETAIL " and, where gpurun_out/ exists, to
gpurun_out/bench_detail.json.

N > 1: the frame is sharded by interleaved 16-row strips over the ranks (strong scaling), the scene is broadcast from
rank 0 over RCCL, every rank accumulates its own bands, and the accumulated frame is gathered to rank 0 with RCCL after
every 16 x N steps -- about 100 presentations per second whatever N is -- and after the last step, inside the timed region
(on a side stream while rendering goes on; the last gather is waited for before the clock stops).  A 4K RGBA32F frame is
132.7 MB: gathered after every 0.1 ms step it would measure the xGMI links.  --dist cabi runs broadcast and gather through
the library's own RCCL calls (include/rt_dist.h) instead of torch.distributed; the control plane (barriers, the timing
reduction) stays on a gloo group either way.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BAND_ROWS = 16
HBM_PEAK_GBS = 8000.0           # MI355X HBM3E peak
VALU_PEAK_TFLOPS_NO_FMA = 78.6  # 157.3 TFLOP/s fp32 vector peak counts an FMA as 2; this path may not fuse
CPU_BASELINE_SECONDS = 12.0     # the CPU leg renders whole frames of the same workload until this much time has passed


def llvmpipe_reference(scene: str, w: int, h: int, limit: int):
    """The reference's own shader on Mesa llvmpipe for this workload, as measured in the build container on 8 threads when the
    whole-frame fixtures were made (tests/golden/ref_fullframe_digests.json: best and mean of three steady-state frames; the
    shader text does not travel to the GPU box, so this leg cannot be re-run there), or None."""
    try:
        with open(os.path.join(ROOT, "tests", "golden", "ref_fullframe_digests.json")) as f:
            entries = json.load(f)
    except (OSError, ValueError):
        return None
    for e in entries.values():
        lim = int(np.array(e["uniforms_bits"][19:20], np.uint32)[0])
        if (e["scene"], e["width"], e["height"], lim) == (scene, w, h, limit):
            ms = [e["llvmpipe"]["ms_best"], e["llvmpipe"]["ms_mean"]]
            return {"ms_per_frame": [round(v, 1) for v in ms], "Mrays/s": [round(e["rays"] / v / 1e3, 2) for v in reversed(ms)], "cores": e["llvmpipe"]["threads"],
                    "where": "build container, not this host (the shader text cannot travel)"}
    return None

SCENE_TEXT = {
    "bunny": "default scene (bundled Stanford bunny for the missing dragon, 69 451 triangles, Octree(18,10,6,6))",
    "synthetic_100k": "synthetic scene of 100 000 primitives (icospheres and boxes, 1.6 M triangles, seed 1, depth-8 octree)",
    "synthetic_1m": "synthetic scene of 1.28 M triangles (1 000 perturbed icospheres, seed 2, depth-18 octree, 8.9 M triangle references)",
}
N_CU = 256                        # compute units of an MI355X
GATHER_CEILING = [2200, 3250]     # lane-loads per us per CU that tools/experiments/gather_bench.hip reaches at 26 / 64 active lanes (64-byte records, L2-resident)


def traversal_bytes(st: dict) -> int:
    """Bytes the reference's data layout moves for the counted octree work: per node visit 2 header words + 4 B per
    child pointer; per triangle test a 16 B index record + three 16 B vertices."""
    return 8 * st["node_pops"] + 4 * st["child_box_tests"] + 64 * st["triangle_tests"]


def frame_bytes(st: dict, pixels: int, blended: bool) -> int:
    """traversal_bytes + per shaded triangle hit index + 3 vertices + 3 normals (112 B) + per pixel a 16 B colour
    write (+16 B read when blending with n > 1)."""
    return traversal_bytes(st) + 112 * st["triangle_hits_shaded"] + pixels * (16 + (16 if blended else 0))


def counted_flops(st: dict) -> int:
    """Arithmetic of the tests that were actually executed: sphere 17 flop, child box 27, triangle 50 (counted from the
    shader text).  The sphere tests are the ones the kernels ran, not the shader's 84 per ray."""
    return 17 * st["sphere_tests"] + 27 * st["child_box_tests"] + 50 * st["triangle_tests"]


def workload_key(scene: str, w: int, h: int, limit: int, traversal: str) -> str:
    return f"{scene}:{w}x{h}:limit{limit}:{traversal}"


def counter_digest(key: str, kernel: str):
    """What the committed rocprofv3 --pmc passes of this very workload say about `kernel` (profiles/pmc_digest.json,
    written by tools/profile_digest.py) and the build id of the library they were collected on, or (None, None) when that
    workload has not been profiled."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_digest.json")) as f:
            w = json.load(f).get(key, {})
        return w.get(kernel), w.get("_build_id")
    except (OSError, ValueError):
        return None, None


def self_launch(args) -> int:
    """--gpus N > 1 without a launcher: run the N ranks as a child process group and relay rank 0's line.  Nothing in this
    process has touched a GPU yet (no HIP call, no torch.cuda call)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "16")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    line = None
    for ln in r.stdout.decode(errors="replace").splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if line is not None:
        print(line, flush=True)
    if r.returncode != 0:
        return r.returncode
    return 0 if line is not None else 1


def raster_pass_timing(width, height, device):
    """The pass that follows the ray-tracing draw in the reference's frame (App.cpp:142-166: nine draws of its mesh, depth-tested
    against the ray tracer's depth), timed on device planes at the benchmark's frame size: an extra of the JSON line, not part of `value`."""
    import torch
    from raytracer_amd import api
    from raytracer_amd.camera import Camera
    assets = os.path.join(ROOT, "tests", "golden", "overlay_assets.npz")
    if not os.path.exists(assets):
        return None
    z = np.load(assets)
    worlds, its = [], []
    for i in (-1, 0, 1):                       # App.cpp:157-162
        for j in (-1, 0, 1):
            t = np.array([4.0 * i, 4.0 + 4.0 * j, 0.0], np.float32)
            w = np.eye(4, dtype=np.float32); w[3, :3] = t
            it = np.eye(4, dtype=np.float32); it[:3, 3] = -t
            worlds.append(w.reshape(16)); its.append(it.reshape(16))
    cam = Camera(eye=(-2.5, 5.25, 11.0), at=(-2.5, 5.25, 4.1), aspect=width / height)
    mesh = api.OverlayMesh(z["vertices"], z["indices"], z["texture"], device=device)
    dev = torch.device("cuda", device)
    col = torch.zeros((height, width, 4), dtype=torch.float32, device=dev); d24 = torch.full((height, width), 0xFFFFFF, dtype=torch.int32, device=dev)
    args_ = (cam.view_proj().ravel(), worlds, its, (-1.0, -2.0, -0.5), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))      # light: App.hpp:79-81
    for _ in range(3):
        mesh.draw(col, d24, *args_)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    calls = 20
    t0.record()
    for _ in range(calls):
        mesh.draw(col, d24, *args_)
    t1.record(); torch.cuda.synchronize()
    covered = int((d24 != 0xFFFFFF).sum())
    mesh.close()
    return {"what": "rt_overlay_draw (App.cpp:147-164), device planes", "ms_per_call": round(t0.elapsed_time(t1) / calls, 4), "triangles": 9 * int(len(z["indices"]) // 3), "covered_pixels": covered}


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
    ap.add_argument("--frames-in-flight", type=int, default=None, help="frames of a sequence traced concurrently (default: the library's choice)")
    ap.add_argument("--per-frame-calls", action="store_true", help="one rt_render per step instead of rt_render_sequence calls")
    ap.add_argument("--present-every", type=int, default=0,
                    help="N > 1: the accumulated frame is gathered to rank 0 after every this many steps and after the last one. "
                         "0 = 16 x N: a step takes about 0.67 ms / N, so the frame is presented about 100 times a second whatever N is "
                         "(a 4K RGBA32F frame is 132.7 MB: gathered after every step it would measure the xGMI links, not the renderer)")
    ap.add_argument("--traversal", default="auto", choices=["auto", "reference", "near_first", "near_first_exact"],
                    help="child order of the octree traversal: auto (default: the library picks between the next two, outputs bit-identical to the reference "
                         "shader's either way); the shader's own; nearest first with the shader's result (RT_ORDER_NEAR_FIRST_EXACT in include/rt_hotpath.h); "
                         "nearest first (colour tolerance only)")
    ap.add_argument("--dist", default="torch", choices=["torch", "cabi"], help="N > 1 data plane: torch.distributed (RCCL) or the library's own RCCL calls (include/rt_dist.h)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-north-star", action="store_true", help="skip the sub-reports of the two deep scenes and the raster pass")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(self_launch(args))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s)")

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

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the ray-tracing pass has no CPU fallback")
    # RT_BENCH_REHEARSE=1: every rank on GPU 0 with the gloo backend -- a functional rehearsal of the N > 1 path on a
    # one-GPU box (RCCL does not accept two ranks on one device); never a measurement
    rehearse = os.environ.get("RT_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # RT_BENCH_FORCE_DIST=1 (with torch.distributed.run --nproc-per-node 1): the N > 1 code path -- process group,
    # scene broadcast, pipelined gather -- with a single rank, to exercise the RCCL calls on a one-GPU box
    use_dist = world > 1 or os.environ.get("RT_BENCH_FORCE_DIST") == "1"
    cabi = use_dist and args.dist == "cabi"
    if cabi and rehearse:
        raise SystemExit("--dist cabi needs one GPU per rank (RCCL does not accept two ranks on one device)")
    backend = None
    comm = None
    if use_dist:
        if rehearse or cabi:
            dist.init_process_group("gloo")     # (with --dist cabi: the control plane only)
        else:
            dist.init_process_group("nccl", device_id=device)
        backend = dist.get_backend()
        assert dist.get_world_size() == world
        if cabi:
            from raytracer_amd import dist_cabi
            box = [dist_cabi.make_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            comm = dist_cabi.DistComm(box[0], world, rank, local_rank)

    W, H = args.width, args.height
    parts = world
    stream = torch.cuda.current_stream().cuda_stream
    present_every = args.present_every if args.present_every > 0 else 16 * max(world, 1)
    sequence = not args.per_frame_calls
    detail = {}

    def progress(msg):
        """RT_BENCH_PROGRESS=1: phase marks of every rank on stderr (to see where a multi-rank run stands)."""
        if os.environ.get("RT_BENCH_PROGRESS") == "1":
            print(f"[bench rank {rank}] {msg}", file=sys.stderr, flush=True)

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def run_workload(tag: str, scene_name: str, limit: int, steps: int, warmup: int, cpu_leg: bool, traversal: str) -> dict:
        """Builds the scene (rank 0, device octree builder; broadcast over RCCL), warms up, times `steps` frames between
        barriers, counts their work in an untimed replay, and returns the short report of this workload (the long one goes to `detail[tag]`)."""
        t_build = time.time()
        scene_arrays = None
        if rank == 0:
            oc = {"bunny": scenes.default_octree, "synthetic_100k": scenes.synthetic_100k, "synthetic_1m": scenes.synthetic_1m}[scene_name](device=local_rank)
            scene_arrays = oc.as_dict()
            scene_arrays["sky"] = scenes.procedural_skybox(900)
        if cabi:
            sc = comm.scene_create(scene_arrays)
            comm.skybox_set(sc, scene_arrays["sky"] if rank == 0 else None)
            sky = scene_arrays["sky"] if rank == 0 else None
        else:
            if use_dist:
                scene_arrays = rdist.broadcast_scene(scene_arrays, device)
            sky = scene_arrays["sky"]
            sc = api.Scene.from_dict(scene_arrays, device=local_rank)
            sc.set_skybox(sky)
        sc.set_pipeline(args.pipeline)
        sc.set_traversal_order(traversal)
        if args.subframes:
            sc.set_subframes(args.subframes)
        if args.frames_in_flight:
            sc.set_frames_in_flight(args.frames_in_flight)
        t_build = time.time() - t_build

        cam = Camera(aspect=W / H)

        def uniforms(step):
            return Uniforms.from_camera(cam, time=step / 60.0, max_recursion_limit=limit).pack()

        compact = parts > 1 or cabi      # (the library's own gather takes compact band buffers, also from a single rank)
        fr0 = api.make_frame(uniforms(0), W, H, band=(BAND_ROWS, parts, rank), compact=compact)
        rows_local = api.band_row_count(fr0) if compact else H
        rows_pad = rdist.padded_row_count(H, BAND_ROWS, parts) if compact else H
        color = torch.zeros((rows_pad, W, 4), dtype=torch.float32, device=device)

        def frame_of(step, accum_index):
            return api.make_frame(uniforms(step), W, H, accum_index=accum_index, blend=True, band=(BAND_ROWS, parts, rank), stream=stream, compact=compact)

        def target_of(sync=False):
            tg = api.rt_target()
            tg.memory, tg.sync, tg.color = api.RT_MEM_DEVICE, int(sync), color.data_ptr()
            return tg

        def render(step, accum_index, sync=False):
            sc.render_raw(frame_of(step, accum_index), target_of(sync))

        gatherer = rdist.FrameGatherer(H, BAND_ROWS) if use_dist and not cabi else None
        full = torch.zeros((H, W, 4), dtype=torch.float32, device=device) if cabi and rank == 0 else None

        def present():
            """The accumulated frame's bands to rank 0 (enqueued; rendering goes on)."""
            if cabi:
                comm.gather_frame(fr0_stream, color.data_ptr(), full.data_ptr() if full is not None else None, 16, stream)
            else:
                gatherer.submit(color)
        fr0_stream = frame_of(0, 1)

        # One step = one frame.  Frames are issued as the library's progressive-accumulation sequences (rt_render_sequence).
        # On several GPUs every rank accumulates its own bands; the accumulated frame is presented -- its bands gathered to
        # rank 0 while rendering goes on -- after every `present_every` steps and after the last one.
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
                    present()
            if use_dist and not cabi:
                gatherer.last()

        # the per-frame inputs (camera matrices, time) are prepared before the clock starts: they are the caller's data
        warm_frames = [frame_of(k, k + 1) for k in range(warmup)]
        timed_frames = [frame_of(k, k + 1) for k in range(steps)]

        # untimed: lets the library's selection of pipeline and child order settle on this workload (it times its own first frames)
        progress(f"{tag}: scene ready")
        if args.pipeline == "auto" or traversal == "auto":
            for k in range(16):
                if sequence:
                    sc.render_sequence_raw([frame_of(4 * k + j, j + 1) for j in range(4)], target_of(sync=True))
                else:
                    render(k, k + 1, sync=True)
        progress(f"{tag}: selection settled on {sc.last_mode()}")
        if warmup:
            run_steps(warm_frames)
        progress(f"{tag}: warm-up enqueued")
        sync_all()
        progress(f"{tag}: warm-up done")
        _, _, cand_ms = sc.selection_state()

        # the dominant kernel's launches are bracketed with HIP events on their own streams inside the library
        sc.set_kernel_timing(True)
        t0 = time.perf_counter()
        run_steps(timed_frames)
        progress(f"{tag}: timed steps enqueued")
        sync_all()
        progress(f"{tag}: timed steps done")
        elapsed = time.perf_counter() - t0
        k_total_ms, k_launches = sc.kernel_time()
        sc.set_kernel_timing(False)
        pipeline, order = sc.last_mode()         # what the timed frames actually ran with
        if traversal != "auto" and order != traversal and not (traversal == "near_first_exact" and pipeline == "megakernel"):
            raise SystemExit(f"{tag}: asked for child order {traversal}, the library ran {order} (a scene with too many face-coplanar triangles?)")
        if use_dist:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        # the accumulated frame the K steps leave (on rank 0: gathered), as a checksum
        if use_dist:
            shown = full if cabi else gatherer.last()
        else:
            shown = color
        crc = zlib.crc32(shown.cpu().numpy().tobytes()) if (rank == 0 and shown is not None) else None
        progress(f"{tag}: frame checksum taken")

        # from here on the scene is pinned to what was timed (left alone, the selection looks at the other candidates now and then)
        sc.set_pipeline(pipeline)
        sc.set_traversal_order(order)

        # ---- wavefront pipeline: the traversal kernel's own duration, with the sub-frames serialised (in the timed frames several
        # sub-frame chains overlap on separate streams, so a launch's event pair also spans time it shared the machine) -----------
        serial = None
        if pipeline == "wavefront":
            sc.set_subframes(1); sc.set_frames_in_flight(1)
            n_ser = min(steps, 4)
            ser_frames = [frame_of(k, k + 1) for k in range(n_ser)]
            sc.render_sequence_raw(ser_frames[:2], target_of(sync=True))
            torch.cuda.synchronize()       # (no barrier here: the ranks choose their pipelines independently, and only those on the wavefront pipeline come this way)
            sc.set_kernel_timing(True)
            t1 = time.perf_counter()
            sc.render_sequence_raw(ser_frames, target_of(sync=True))
            torch.cuda.synchronize()
            ser_elapsed = time.perf_counter() - t1
            s_ms, s_n = sc.kernel_time()
            sc.set_kernel_timing(False)
            sc.set_subframes(args.subframes or 0); sc.set_frames_in_flight(args.frames_in_flight or 0)
            serial = {"frame_ms": 1e3 * ser_elapsed / n_ser, "kernel_ms": s_ms / max(s_n, 1), "launches_per_frame": s_n / n_ser}

        # ---- count the work of the timed frames (untimed replay with counters on) ------------------------------
        sc.set_counting(True)
        tot = {}
        for k in range(steps):
            render(k, k + 1, sync=True)
            for name, v in sc.stats().items():
                tot[name] = tot.get(name, 0) + v
        sc.set_counting(False)
        local_tot = dict(tot)
        if use_dist:
            keys = sorted(tot)
            t = torch.tensor([tot[k] for k in keys], dtype=torch.int64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(t)
            tot = {k: int(v) for k, v in zip(keys, t.tolist())}

        rays = tot["rays"]
        frame_ms = 1e3 * elapsed / steps
        segs = "primary segment only" if limit == 1 else f"primary + up to {limit - 1} secondary segment{'s' if limit > 2 else ''}"
        rep = {
            "value": round(rays / elapsed / 1e6, 3), "ms_per_step": round(frame_ms, 4), "unit": "Mrays/s", "steps": steps, "warmup": warmup,
            "config": {"workload": f"{SCENE_TEXT[scene_name]} at {W}x{H}, max_recursion_limit={limit} ({segs}), 1 spp/frame, progressive accumulation",
                       "scene": scene_name, "width": W, "height": H, "max_recursion_limit": limit,
                       "pipeline": pipeline, "traversal_order": order,
                       "selected_by": "library" if (args.pipeline == "auto" or traversal == "auto") else "caller",
                       "candidates_ms": {k: round(v, 3) for k, v in cand_ms.items() if v is not None and v < 1e29},
                       "parallelism": f"interleaved {BAND_ROWS}-row bands x{world}" + (f", gather every {present_every} steps ({'rt_dist.h' if cabi else 'torch.distributed'})" if world > 1 else "")},
            "rays_per_frame": round(rays / steps), "frame_crc32": crc,
        }
        # ---- roofline of the dominant kernel, this rank's launches ----------------------------------------------
        kernel = {"wavefront": "traverse_kernel", "megakernel": "render_kernel", "persistent": "path_kernel"}[pipeline]
        if pipeline == "megakernel" and k_launches < steps:
            kernel = "render_batch_kernel"   # the library traced several frames of the sequence per launch
        k_ms = k_total_ms / max(k_launches, 1)
        px_local = W * min(rows_local, H) if compact else W * H
        all_bytes = frame_bytes(local_tot, px_local * steps, blended=True)
        if pipeline == "wavefront":
            # several kernels per frame, sub-frames overlapping: the frame's algorithmic bytes over the frame's time
            basis = "frame"
            achieved = all_bytes / steps / (frame_ms * 1e-3) / 1e9
            abytes = all_bytes / steps
        else:
            basis = "launch"
            abytes = all_bytes / max(k_launches, 1)
            achieved = abytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        pmc, pmc_build = counter_digest(workload_key(scene_name, W, H, limit, order), kernel)
        if pmc and not all(k in pmc for k in ("kernel_ms", "valu_issue_frac", "wait_frac")):
            pmc = None      # (a kernel the profiled run of this workload did not launch)
        verdict_only = None
        if pmc and world > 1:
            # the counters were collected on one GPU; with the frame split over several the same kernel runs on a part of it: only the
            # verdict on what bounds the kernel is carried over, not the per-launch figures
            verdict_only, pmc = pmc, None
        bound = "hbm"
        if verdict_only:
            bound = verdict_only.get("bound_one_gpu", "unprofiled") + " (one-GPU profile)"
        load_path = None
        if pmc:
            hbm_frac = pmc["hbm_bytes_per_launch"] / (pmc["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            cands = {"hbm": hbm_frac, "valu-issue": pmc["valu_issue_frac"]}
            if kernel == "traverse_kernel" and "lane_loads_per_launch" in pmc and frame_ms > 0:
                # scattered record fetches that miss the vector L1: measured against what the load path sustains for such fetches
                per_frame = k_launches / steps
                load_path = {"lane_loads_per_us_per_cu": round(pmc["lane_loads_per_launch"] * per_frame / (frame_ms * 1e3) / N_CU, 1),
                             "l2_requests_per_us_per_cu": round(pmc.get("l2_requests_per_launch", 0.0) * per_frame / (frame_ms * 1e3) / N_CU, 1),
                             "gather_ceiling_lane_loads_per_us_per_cu": GATHER_CEILING}
                cands["load-path"] = load_path["lane_loads_per_us_per_cu"] / GATHER_CEILING[0]
            top = max(cands, key=cands.get)
            bound = top if cands[top] >= 0.5 else f"latency ({top} {cands[top]:.2f} nearest; waves wait {pmc['wait_frac']:.2f})"
        stale = None if pmc_build is None and not pmc and not verdict_only else (pmc_build != api.build_id())
        rep["roofline"] = {"kernel": kernel, "bound": bound, "basis": basis, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc["hbm_bytes_per_launch"] if pmc else None,
                           "algorithmic_bytes": int(abytes), "kernel_ms": round(k_ms, 4), "launches_per_frame": round(k_launches / steps, 3),
                           "overlapped": bool(k_ms * k_launches / steps > frame_ms * 1.001), "counters_stale": stale}
        if serial:
            # per launch of the traversal kernel alone, from the serialised frames: its share of the counted bytes over its own duration
            t_bytes = traversal_bytes(local_tot) / steps / max(serial["launches_per_frame"], 1e-9)
            rep["roofline"]["kernel_serialised"] = {"frame_ms": round(serial["frame_ms"], 3), "kernel_ms": round(serial["kernel_ms"], 4), "launches_per_frame": round(serial["launches_per_frame"], 3),
                                                    "achieved": round(t_bytes / (serial["kernel_ms"] * 1e-3) / 1e9, 1), "frac": round(t_bytes / (serial["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        if pmc:
            rep["roofline"]["counters"] = {k: (round(pmc[k], 3) if isinstance(pmc[k], float) else pmc[k]) for k in ("valu_issue_frac", "lanes_active_of_64", "occupancy_waves_per_simd", "wait_frac", "l2_hit_rate", "hbm_gbs") if k in pmc}
        aflops = counted_flops(local_tot) / steps
        rep["valu"] = {"achieved_tflops": round(aflops / (frame_ms * 1e-3) / 1e12, 3), "peak_tflops_no_fma": VALU_PEAK_TFLOPS_NO_FMA,
                       "frac": round(aflops / (frame_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS_NO_FMA, 4)}
        detail[tag] = {"work_per_frame": {k: v / steps for k, v in tot.items()}, "scene_build_s": round(t_build, 3), "counters": pmc, "load_path": load_path,
                       "counters_collected_on_build": pmc_build, "issue": ("rt_render_sequence over the K steps" if world == 1 else f"rt_render_sequence per {present_every} steps") if sequence else "one rt_render per step",
                       "roofline_what": "achieved = bytes the REFERENCE's data layout would move for the counted node visits, triangle tests, shaded hits and pixel writes / duration: per launch "
                                        "of the one kernel (basis launch), per frame over the frame time where several overlapping kernels make a frame (basis frame); an effective rate -- traffic "
                                        "and counters are the measured side (profiles/pmc_digest.json of the same workload; counters_stale: collected on another build of the kernels)",
                       "valu_what": "flop of the sphere, box and triangle tests actually executed (17 / 27 / 50 each) over the frame time",
                       "sphere_tests_per_ray": round(local_tot["sphere_tests"] / max(local_tot["rays"], 1), 2)}

        # ---- CPU baseline: the oracle (a port of the shader with the reference vehicle's numerics) on this host -
        if cpu_leg and rank == 0 and world == 1:
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
            rep["cpu_baseline"] = {"value": round(n_rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
                                   "sample": f"{n_frames} consecutive {W}x{H} frames of the same workload in {dt:.1f} s"}
            ref = llvmpipe_reference(scene_name, W, H, limit)
            if ref:
                rep["cpu_baseline"]["reference_llvmpipe"] = ref
        sc.close()
        del color
        return rep

    main_rep = run_workload("main", args.scene, args.max_recursion_limit, args.steps, args.warmup, cpu_leg=not args.no_cpu_baseline, traversal=args.traversal)
    out = {
        "metric": "Mrays/s (primary+secondary) & frame ms at 4K, 1/2/4/8 MI355X vs llvmpipe CPU",
        "value": main_rep.pop("value"), "unit": main_rep.pop("unit"), "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": main_rep.pop("ms_per_step"), "higher_is_better": True, "scaling": "strong" if world > 1 else "weak",
        "vs_baseline": None, "dtype": "f32",
        "data": ("REHEARSAL on one GPU, not a measurement; " if rehearse else "") + "synthetic camera/skybox; bundled bunny mesh (the default scene's dragon is absent from the reference checkout)",
        "ranks": {"world_size": dist.get_world_size() if use_dist else 1, "backend": ("rt_dist.h (RCCL) + gloo control plane" if cabi else backend + (" (RCCL)" if backend == "nccl" else "")) if backend else "none (single process)"},
        "build_id": api.build_id(),
    }
    main_rep.pop("steps"); main_rep.pop("warmup")
    out.update(main_rep)
    if not args.no_north_star:
        subs = (("north_star", "synthetic_1m", "auto"), ("north_star_reference_order", "synthetic_1m", "reference"),
                ("config3", "synthetic_100k", "auto"), ("config3_reference_order", "synthetic_100k", "reference"))
        for name, scene_name, order in subs:
            if (args.scene, args.max_recursion_limit) == (scene_name, 5):
                continue
            ns = run_workload(name, scene_name, 5, max(4, min(args.steps, 20)), min(args.warmup, 2), cpu_leg=False, traversal=order)
            cfg = ns.pop("config")
            ns.update({k: cfg[k] for k in ("scene", "pipeline", "traversal_order", "selected_by", "candidates_ms")})
            ns.pop("valu", None)
            ref = llvmpipe_reference(scene_name, W, H, 5)
            if ref:      # (the same workload on the reference's own vehicle: frame times measured in the build container)
                ns["reference_llvmpipe_ms_per_frame"] = ref["ms_per_frame"]
            for k in ("peak", "unit", "algorithmic_bytes"):      # (as in the headline report; the line has to stay short)
                ns["roofline"].pop(k, None)
            out[name] = ns
        for a, b in (("north_star", "north_star_reference_order"), ("config3", "config3_reference_order")):
            if a in out and b in out and out[a]["frame_crc32"] is not None:
                out[a]["frame_identical_to_reference_order"] = out[a]["frame_crc32"] == out[b]["frame_crc32"]
        if rank == 0:
            out["raster_pass"] = raster_pass_timing(args.width, args.height, local_rank)
    if rank == 0:
        os.write(result_fd, (json.dumps(out, separators=(",", ":")) + "\n").encode())
        blob = json.dumps(detail)
        sys.stderr.write("BENCH_DETAIL " + blob + "\n")
        side = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(side) and os.access(side, os.W_OK):
            with open(os.path.join(side, "bench_detail.json"), "w") as f:
                f.write(blob)
    if comm is not None:
        comm.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
