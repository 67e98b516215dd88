#!/usr/bin/env python3
"""Time one rank's share of a 4K frame for several world sizes on one GPU (development aid): part 0 of N interleaved
16-row bands (RT_TIME_BANDS_ROWS), compact target, issued as rt_render_sequence calls of 16 (RT_TIME_BANDS_SEQ) frames.
usage: tools/time_bands.py [bunny|s100k|s1m] [max_recursion_limit]
environment: RT_TIME_BANDS_SIZE=WxH, RT_TIME_BANDS_PARTS=1,2,4,8, RT_TIME_BANDS_CHAINS=<sub-frames>,<frames in flight>,
RT_TIME_BANDS_STACK=<frames per launch chain>, RT_TIME_BANDS_BATCH=<frames per launch of the one-kernel pipeline> (0: the library's choice)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import api, scenes, dist as rdist
from raytracer_amd.camera import Camera
from raytracer_amd.params import Uniforms
W, H = (int(v) for v in os.environ.get("RT_TIME_BANDS_SIZE", "3840x2160").split("x"))
MAX_BATCH = int(os.environ.get("RT_TIME_BANDS_BATCH", "0"))      # 0: the library's choice
ROWS = int(os.environ.get("RT_TIME_BANDS_ROWS", "16"))           # strip height
SEQ = int(os.environ.get("RT_TIME_BANDS_SEQ", "16"))             # frames per rt_render_sequence call
NAME = sys.argv[1] if len(sys.argv) > 1 else "bunny"
LIMIT = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sc = api.Scene.from_dict({"bunny": scenes.default_octree, "s100k": scenes.synthetic_100k, "s1m": scenes.synthetic_1m}[NAME](device=0).as_dict()); sc.set_skybox(scenes.procedural_skybox(900))
sc.set_sequence_batch(MAX_BATCH)
if os.environ.get("RT_TIME_BANDS_CHAINS"):      # "sub-frames,frames in flight"
    k, fl = map(int, os.environ["RT_TIME_BANDS_CHAINS"].split(",")); sc.set_subframes(k); sc.set_frames_in_flight(fl)
sc.set_frames_per_launch(int(os.environ.get("RT_TIME_BANDS_STACK", "0")))
for parts in (int(v) for v in os.environ.get("RT_TIME_BANDS_PARTS", "1,2,4,8").split(",")):
    rows = rdist.padded_row_count(H, ROWS, parts) if parts > 1 else H
    color = torch.zeros((rows, W, 4), dtype=torch.float32, device="cuda")
    frames = [api.make_frame(Uniforms.from_camera(Camera(aspect=W / H), time=k / 60.0, max_recursion_limit=LIMIT).pack(), W, H, accum_index=k + 1, band=(ROWS, parts, 0), compact=parts > 1,
                             stream=torch.cuda.current_stream().cuda_stream) for k in range(4 * SEQ)]
    tg = api.rt_target(); tg.memory, tg.sync, tg.color = api.RT_MEM_DEVICE, 1, color.data_ptr()
    for k in range(16):      # untimed, waited for: the library's selection of pipeline and child order settles (it times its own frames)
        sc.render_sequence_raw(frames[:4], tg)
    tg.sync = 0
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for first in range(0, 4 * SEQ, SEQ):
            sc.render_sequence_raw(frames[first:first + SEQ], tg)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / (4 * SEQ) * 1e3
    print(f"parts={parts}: {ms:.4f} ms per frame share  -> ideal {ms * parts:.3f} ms x 1/{parts}, mode {sc.last_mode()}", flush=True)
