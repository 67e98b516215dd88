#!/usr/bin/env python3
"""The camera-path workload: 7680x4320, 64 accumulated frames per pose, 120 poses on a fixed orbit, bands over the ranks.

  python tools/run_orbit.py [--poses 120] [--spp 64] [--width 7680 --height 4320] [--max-recursion-limit 5]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/run_orbit.py ...

Every rank accumulates its own interleaved 16-row bands; a finished pose is gathered to rank 0 once (RCCL), beside the
rendering of the next pose.  Rank 0 prints one JSON line: ms per pose, frames/s, Mrays/s.  Rays are counted on an untimed
replay of the FIRST frame of every pose with the work counters on, times spp (frames of a pose differ only in their
random seeds)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--poses", type=int, default=120, help="number of poses rendered (from pose 0 on)")
    ap.add_argument("--n-poses", type=int, default=120, help="poses on the full circle")
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--width", type=int, default=7680)
    ap.add_argument("--height", type=int, default=4320)
    ap.add_argument("--max-recursion-limit", type=int, default=5)
    ap.add_argument("--scene", default="bunny", choices=["bunny", "synthetic_100k", "synthetic_1m"])
    ap.add_argument("--checksum", action="store_true", help="rank 0 copies every presented frame to the host and prints its CRC32")
    args = ap.parse_args()
    sys.stdout.flush()
    result_fd = os.dup(1)      # stdout carries the one JSON line; RCCL's start-up banner and the like go to stderr
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from raytracer_amd import api, scenes, orbit, dist as rdist

    world, rank, local_rank = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU: the ray-tracing pass has no CPU fallback")
    rehearse = os.environ.get("RT_BENCH_REHEARSE") == "1"      # all ranks on GPU 0 over gloo: functional rehearsal only
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("gloo") if rehearse else dist.init_process_group("nccl", device_id=device)

    arrays = None
    if rank == 0:
        arrays = {"bunny": scenes.default_octree, "synthetic_100k": scenes.synthetic_100k, "synthetic_1m": scenes.synthetic_1m}[args.scene](device=local_rank).as_dict()
        arrays["sky"] = scenes.procedural_skybox(900)
    if world > 1:
        arrays = rdist.broadcast_scene(arrays, device)
    sc = api.Scene.from_dict(arrays, device=local_rank)
    sc.set_skybox(arrays["sky"])

    cfg = orbit.OrbitConfig(width=args.width, height=args.height, spp=args.spp, n_poses=args.n_poses, max_recursion_limit=args.max_recursion_limit)
    stream = torch.cuda.current_stream().cuda_stream
    drv = orbit.OrbitDriver(cfg, orbit.scene_pose_renderer(sc, cfg, world=world, rank=rank, stream=stream), device, world=world, rank=rank)
    sums = []
    sink = (lambda pose, frame: sums.append(zlib.crc32(frame.cpu().numpy().tobytes()))) if args.checksum else None

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    drv.run(range(min(2, args.poses)))            # untimed: the library settles its pipeline choice and workgroup order
    sync_all()
    sums.clear()
    t0 = time.perf_counter()
    drv.run(range(args.poses), sink=sink)
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    pipeline, _ = sc.pipeline_state()
    sc.set_counting(True)
    rays = 0
    scratch = torch.zeros_like(drv.color)
    for pose in range(args.poses):
        u = orbit.pose_uniforms(orbit.OrbitConfig(**{**cfg.__dict__, "spp": 1}), pose)[0]
        fr = api.make_frame(u, cfg.width, cfg.height, band=(cfg.band_rows, world, rank), compact=world > 1)
        tg = api.rt_target()
        tg.memory, tg.sync, tg.color = api.RT_MEM_DEVICE, 1, scratch.data_ptr()
        sc.render_raw(fr, tg)
        rays += sc.stats()["rays"] * cfg.spp
    sc.set_counting(False)
    if world > 1:
        t = torch.tensor([rays], dtype=torch.int64, device=device)
        dist.all_reduce(t)
        rays = int(t.item())
    if rank == 0:
        out = {"workload": f"{args.scene} {cfg.width}x{cfg.height}, {cfg.spp} accumulated frames x {args.poses} poses of a {cfg.n_poses}-pose orbit, max_recursion_limit={cfg.max_recursion_limit}",
               "n_gpus": world, "rehearsal": rehearse, "seconds": round(elapsed, 4), "ms_per_pose": round(1e3 * elapsed / args.poses, 3),
               "ms_per_frame": round(1e3 * elapsed / (args.poses * cfg.spp), 4), "frames_per_s": round(args.poses * cfg.spp / elapsed, 2),
               "Mrays_per_s": round(rays / elapsed / 1e6, 1), "rays": rays, "pipeline": pipeline,
               "presentation": "bands gathered to rank 0 once per pose on a side stream" if world > 1 else "frame stays on the device"}
        if args.checksum:
            out["crc32_of_frames"] = sums
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
