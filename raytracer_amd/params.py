"""The per-frame uniform block of the ray-tracing pass, as the reference sets it.

Field-for-field the uniforms of /root/reference/app/assets/ray_tracer.frag:25-49 that the
reference's App::Render uploads at /root/reference/app/core/src/App.cpp:120-138 (the shader's
``width``/``height`` uniforms are unused by the shader and are carried as the integer frame size
instead; the octree bounds belong to the scene object).  ``pack()`` produces the 70-word
little-endian block that is the memory image of ``rt_uniforms`` in include/rt_hotpath.h.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from .camera import Camera, Portals

f32 = np.float32
N_UNIFORM_WORDS = 70


@dataclass
class Uniforms:
    inv_view_proj_mat: np.ndarray
    camera_position: tuple
    max_recursion_limit: int = 5          # App.cpp:126
    time: float = 0.0                     # App.cpp:127
    blur_amount: float = 0.00001          # App.cpp:128
    portals: Portals = field(default_factory=Portals)
    z_near: float = 0.01
    z_far: float = 1000.0

    @staticmethod
    def from_camera(cam: Camera, **kw) -> "Uniforms":
        return Uniforms(inv_view_proj_mat=cam.inv_view_proj(), camera_position=tuple(cam.eye), z_near=cam.z_near, z_far=cam.z_far, **kw)

    def pack(self) -> np.ndarray:
        u = np.zeros(N_UNIFORM_WORDS, dtype=f32)
        u[0:16] = np.asarray(self.inv_view_proj_mat, dtype=f32).ravel()
        u[16:19] = np.asarray(self.camera_position, dtype=f32)
        u[19:20].view(np.uint32)[0] = np.uint32(self.max_recursion_limit)
        u[20] = f32(self.time)
        u[21] = f32(self.blur_amount)
        p = self.portals
        u[22:25] = np.asarray(p.position_1, dtype=f32)
        u[25:28] = np.asarray(p.direction_1, dtype=f32)
        u[28:31] = np.asarray(p.position_2, dtype=f32)
        u[31:34] = np.asarray(p.direction_2, dtype=f32)
        u[34] = f32(p.width)
        u[35] = f32(p.height)
        u[36:52] = p.m_1_to_2().ravel()
        u[52:68] = p.m_2_to_1().ravel()
        u[68] = f32(self.z_near)
        u[69] = f32(self.z_far)
        return u
