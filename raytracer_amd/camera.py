"""Host-side producers of the per-frame uniforms of the ray-tracing pass.

These play the role of the reference's Camera (/root/reference/app/core/src/Camera.cpp:5-47,
include/Camera.hpp:46-50) and Portal::GetDifferenceMatrixTo (src/Portal.cpp:15-26): they
only produce *inputs* of the render pass (inv_view_proj_mat, camera_position, the two portal
matrices).  At the drop-in boundary these are opaque float32 values, so the formulas follow glm's
conventions (right-handed, [-1, 1] clip depth, column-major storage) without claiming bit parity
with glm itself.  All matrices are returned as float32 arrays of shape (4, 4) indexed [column][row],
i.e. ``m.ravel()`` is what ``glm::value_ptr`` hands to ``glUniformMatrix4fv``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

f32 = np.float32


def _v3(v) -> np.ndarray:
    return np.asarray(v, dtype=np.float64).reshape(3)


def normalize(v) -> np.ndarray:
    v = _v3(v)
    return v / math.sqrt(float(v @ v))


def perspective(fovy: float, aspect: float, z_near: float, z_far: float) -> np.ndarray:
    """glm::perspective (RH, NO) as used by Camera.cpp:7,26."""
    t = math.tan(fovy / 2.0)
    m = np.zeros((4, 4), dtype=np.float64)
    m[0][0] = 1.0 / (aspect * t)
    m[1][1] = 1.0 / t
    m[2][2] = -(z_far + z_near) / (z_far - z_near)
    m[2][3] = -1.0
    m[3][2] = -(2.0 * z_far * z_near) / (z_far - z_near)
    return m.astype(f32)


def look_at(eye, at, up) -> np.ndarray:
    """glm::lookAt (RH) as used by Camera.cpp:17."""
    eye, at, up = _v3(eye), _v3(at), _v3(up)
    f = normalize(at - eye)
    s = normalize(np.cross(f, up))
    u = np.cross(s, f)
    m = np.eye(4, dtype=np.float64)
    m[0][0], m[1][0], m[2][0] = s
    m[0][1], m[1][1], m[2][1] = u
    m[0][2], m[1][2], m[2][2] = -f
    m[3][0] = -float(s @ eye)
    m[3][1] = -float(u @ eye)
    m[3][2] = float(f @ eye)
    return m.astype(f32)


def matmul_cm(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Product a*b of two [column][row] matrices (glm operator*)."""
    return (b.astype(np.float64) @ a.astype(np.float64)).astype(f32)


def inverse_cm(m: np.ndarray) -> np.ndarray:
    """glm::inverse of a [column][row] matrix (App.cpp:120)."""
    return np.linalg.inv(m.astype(np.float64).T).T.astype(f32)


def rotate(angle: float, axis) -> np.ndarray:
    """glm::rotate(angle, axis) for a unit axis (Portal.cpp:22)."""
    a = normalize(axis)
    c, s = math.cos(angle), math.sin(angle)
    t = (1.0 - c) * a
    m = np.eye(4, dtype=np.float64)
    m[0][0] = c + t[0] * a[0]
    m[0][1] = t[0] * a[1] + s * a[2]
    m[0][2] = t[0] * a[2] - s * a[1]
    m[1][0] = t[1] * a[0] - s * a[2]
    m[1][1] = c + t[1] * a[1]
    m[1][2] = t[1] * a[2] + s * a[0]
    m[2][0] = t[2] * a[0] + s * a[1]
    m[2][1] = t[2] * a[1] - s * a[0]
    m[2][2] = c + t[2] * a[2]
    return m.astype(f32)


def portal_difference_matrix(src_dir, dst_dir) -> np.ndarray:
    """Portal::GetDifferenceMatrixTo (Portal.cpp:15-26)."""
    a, b = _v3(src_dir), _v3(dst_dir)
    axis = np.cross(a, b)
    if math.sqrt(float(axis @ axis)) > 0.0:
        angle = math.acos(max(-1.0, min(1.0, float(a @ b))))
        return rotate(angle, axis)
    return np.eye(4, dtype=f32)


@dataclass
class Camera:
    """Mirror of the reference Camera defaults (Camera.hpp:46-50) plus App.cpp:58-61."""

    eye: tuple = (2.0, 2.0, 2.0)
    at: tuple = (0.0, 0.0, 0.0)
    up: tuple = (0.0, 1.0, 0.0)
    fovy: float = math.radians(90.0)
    aspect: float = 1.0
    z_near: float = 0.01
    z_far: float = 1000.0

    def view_proj(self) -> np.ndarray:
        return matmul_cm(perspective(self.fovy, self.aspect, self.z_near, self.z_far), look_at(self.eye, self.at, self.up))

    def inv_view_proj(self) -> np.ndarray:
        return inverse_cm(self.view_proj())


@dataclass
class Portals:
    """Portal constants of the default scene (App.cpp:31-34)."""

    position_1: tuple = (7.0, 2.5, 1.0)
    direction_1: tuple = field(default_factory=lambda: tuple(normalize((2.0, 0.0, 0.5)).astype(f32).tolist()))
    position_2: tuple = (7.0, 1.5, 5.0)
    direction_2: tuple = (0.0, 0.0, 1.0)
    width: float = 2.0
    height: float = 3.0

    def m_1_to_2(self) -> np.ndarray:
        return portal_difference_matrix(self.direction_1, self.direction_2)

    def m_2_to_1(self) -> np.ndarray:
        return portal_difference_matrix(self.direction_2, self.direction_1)
