"""Headless camera-path driver: the role App plays around the ray-tracing pass, without a window.

The reference's App advances `time` every frame (/root/reference/app/core/src/App.cpp:84-88), restarts the still-frame
counter whenever the camera moves (App.cpp:90-92,209) and otherwise blends frame n into the colour target with weight
1/n (App.cpp:102-105).  This driver replays that for a fixed path: `n_poses` camera poses on a circle around the origin,
eye = radius * (cos a, eye_height, sin a) looking at the origin, and at every pose `spp` accumulated frames with
time_k = k / 60.  With several ranks (one process per GPU, torch.distributed) every rank accumulates its own interleaved
row bands and the finished pose is gathered to rank 0 once, on a side stream, while the next pose is already rendering.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch

from . import dist as rdist
from .camera import Camera
from .params import Uniforms


@dataclass
class OrbitConfig:
    width: int = 7680
    height: int = 4320
    spp: int = 64                  # accumulated frames per pose
    n_poses: int = 120             # poses on the full circle
    max_recursion_limit: int = 5
    radius: float = 3.0
    eye_height: float = 0.7        # eye = radius * (cos a, eye_height, sin a)
    band_rows: int = 16


def orbit_camera(cfg: OrbitConfig, pose: int) -> Camera:
    a = 2.0 * math.pi * (pose % cfg.n_poses) / cfg.n_poses
    eye = (cfg.radius * math.cos(a), cfg.radius * cfg.eye_height, cfg.radius * math.sin(a))
    return Camera(eye=eye, at=(0.0, 0.0, 0.0), aspect=cfg.width / cfg.height)


def pose_uniforms(cfg: OrbitConfig, pose: int) -> list:
    """The packed uniform blocks of the `spp` frames of one pose (time restarts at every pose: time_k = k / 60)."""
    cam = orbit_camera(cfg, pose)
    return [Uniforms.from_camera(cam, time=k / 60.0, max_recursion_limit=cfg.max_recursion_limit).pack() for k in range(cfg.spp)]


def scene_pose_renderer(scene, cfg: OrbitConfig, *, world: int = 1, rank: int = 0, stream: int | None = None):
    """render_pose(uniform_blocks, color) for OrbitDriver on a raytracer_amd.api.Scene: one rt_render_sequence call per
    pose into the rank's (compact) device colour buffer, accumulation index restarting at 1."""
    from . import api

    def render_pose(uniform_blocks, color: torch.Tensor):
        if not color.is_cuda:
            raise RuntimeError("the ray-tracing pass renders into device memory only: no CPU fallback")
        band = (cfg.band_rows, world, rank)
        frames = [api.make_frame(u, cfg.width, cfg.height, accum_index=k + 1, blend=True, band=band, stream=stream, compact=world > 1)
                  for k, u in enumerate(uniform_blocks)]
        tg = api.rt_target()
        tg.memory, tg.sync, tg.color = api.RT_MEM_DEVICE, 0, color.data_ptr()
        scene.render_sequence_raw(frames, tg)
    return render_pose


class OrbitDriver:
    """Runs poses of the path.  `render_pose(uniform_blocks, color)` must leave the accumulated result of the blocks
    (accumulation index 1..len) in this rank's rows of `color`: [padded band rows, width, 4] float32, compact when
    world > 1, the whole frame otherwise.  `sink(pose, frame)` is called on rank 0 with every finished frame
    ([height, width, 4]; valid until the next call) -- for pose p after pose p+1 has been issued, so that presenting
    overlaps rendering."""

    def __init__(self, cfg: OrbitConfig, render_pose, device, *, world: int = 1, rank: int = 0):
        self.cfg, self.render_pose, self.world, self.rank = cfg, render_pose, world, rank
        rows = rdist.padded_row_count(cfg.height, cfg.band_rows, world) if world > 1 else cfg.height
        self.color = torch.zeros((rows, cfg.width, 4), dtype=torch.float32, device=device)
        self.gatherer = rdist.FrameGatherer(cfg.height, cfg.band_rows) if world > 1 else None

    def _present(self, pose, sink):
        frame = self.gatherer.last() if self.gatherer is not None else self.color[:self.cfg.height]
        if sink is not None and self.rank == 0:
            if frame.is_cuda and self.gatherer is None:
                torch.cuda.current_stream(frame.device).synchronize()
            sink(pose, frame)

    def run(self, poses, sink=None):
        waiting = None
        for pose in poses:
            blocks = pose_uniforms(self.cfg, pose)
            if self.gatherer is None and waiting is not None:
                self._present(waiting, sink)           # one buffer: present before the next pose overwrites it
                waiting = None
            self.render_pose(blocks, self.color)
            if self.gatherer is not None:
                if waiting is not None:
                    self._present(waiting, sink)       # the previous pose's gather ran beside this pose's rendering
                self.gatherer.submit(self.color)
            waiting = pose
        if waiting is not None:
            self._present(waiting, sink)
