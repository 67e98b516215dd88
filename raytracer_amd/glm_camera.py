"""ctypes mirror of include/rt_camera.h: the reference's Camera, Portal and CameraManipulator classes with their
binary32 arithmetic in glm's operation order, computed by librt_hotpath.so (host code, no GPU needed).

Reference classes mirrored, with the reference's method names:
  Camera             /root/reference/app/core/src/Camera.cpp:5-47, include/Camera.hpp:11-30
  Portal             /root/reference/app/core/src/Portal.cpp:3-58
  CameraManipulator  /root/reference/app/core/src/CameraManipulator.cpp:11-210 (SDL events as plain numbers)
and `frame_uniforms`, the camera / portal part of App::Render's uniform upload (App.cpp:120-138).

`raytracer_amd.camera` holds the double-precision formulas of the same matrices; the parity fixtures were made with
those.  This module is the float path a caller replaying the reference's interactive camera would use.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import api

f32 = np.float32
KEY_SHIFT, KEY_W, KEY_S, KEY_A, KEY_D, KEY_E, KEY_Q = range(1, 8)


class rt_camera(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("world_up", C.c_float * 3), ("at", C.c_float * 3), ("view", C.c_float * 16),
                ("z_near", C.c_float), ("z_far", C.c_float), ("angle", C.c_float), ("aspect", C.c_float), ("proj", C.c_float * 16)]


class rt_manipulator(C.Structure):
    _fields_ = [("u", C.c_float), ("v", C.c_float), ("distance", C.c_float), ("center", C.c_float * 3), ("speed", C.c_float),
                ("go_forward", C.c_float), ("go_right", C.c_float), ("go_up", C.c_float),
                ("forward_pressed", C.c_int), ("backward_pressed", C.c_int), ("left_pressed", C.c_int), ("right_pressed", C.c_int),
                ("up_pressed", C.c_int), ("down_pressed", C.c_int),
                ("prev_eye", C.c_float * 3), ("prev_center", C.c_float * 3), ("prev_world_up", C.c_float * 3)]


_bound = False


def _lib():
    global _bound
    L = api.lib()
    if not _bound:
        v3, m4 = C.POINTER(C.c_float), C.POINTER(C.c_float)
        cam, man, uni = C.POINTER(rt_camera), C.POINTER(rt_manipulator), C.POINTER(api.rt_uniforms)
        L.rt_camera_init.argtypes = [cam]
        L.rt_camera_set_view.argtypes = [cam, v3, v3, v3]
        L.rt_camera_set_proj.argtypes = [cam, C.c_float, C.c_float, C.c_float, C.c_float]
        L.rt_camera_view_proj.argtypes = [cam, m4]
        L.rt_camera_inv_view_proj.argtypes = [cam, m4]
        L.rt_vec3_normalize.argtypes = [v3, v3]
        L.rt_portal_difference_matrix.argtypes = [v3, v3, m4]
        L.rt_portal_ray.argtypes = [v3, v3, v3, v3, C.c_float, C.c_float, C.c_float]
        L.rt_portal_ray.restype = C.c_float
        L.rt_manipulator_init.argtypes = [man]
        L.rt_manipulator_set_camera.argtypes = [man, cam]
        L.rt_manipulator_update.argtypes = [man, cam, C.c_float, v3, v3, v3, v3, C.c_float, C.c_float]
        L.rt_manipulator_update.restype = C.c_int
        L.rt_manipulator_key.argtypes = [man, C.c_int, C.c_int, C.c_int]
        L.rt_manipulator_mouse_move.argtypes = [man, C.c_int, C.c_int, C.c_int, C.c_int]
        L.rt_manipulator_mouse_wheel.argtypes = [man, C.c_int]
        L.rt_uniforms_set_camera.argtypes = [uni, cam]
        L.rt_uniforms_set_portals.argtypes = [uni, v3, v3, v3, v3, C.c_float, C.c_float]
        for name in ("rt_vec3_normalize", "rt_camera_init", "rt_camera_set_view", "rt_camera_set_proj", "rt_camera_view_proj", "rt_camera_inv_view_proj", "rt_portal_difference_matrix",
                     "rt_manipulator_init", "rt_manipulator_set_camera", "rt_manipulator_key", "rt_manipulator_mouse_move", "rt_manipulator_mouse_wheel",
                     "rt_uniforms_set_camera", "rt_uniforms_set_portals"):
            getattr(L, name).restype = None
        _bound = True
    return L


def _vec(v):
    return (C.c_float * 3)(*[float(f32(x)) for x in v])


def _mat(buf) -> np.ndarray:
    return np.array(buf, dtype=f32).reshape(4, 4)   # [column][row]


def normalize(v) -> np.ndarray:
    """glm::normalize of a vec3 in float."""
    out = (C.c_float * 3)()
    _lib().rt_vec3_normalize(_vec(v), out)
    return np.array(out, dtype=f32)


class Camera:
    """Camera.hpp:5-56 with its method names; matrices are float32 arrays indexed [column][row]."""

    def __init__(self):
        self.c = rt_camera()
        _lib().rt_camera_init(C.byref(self.c))

    def SetView(self, eye, at, world_up):
        _lib().rt_camera_set_view(C.byref(self.c), _vec(eye), _vec(at), _vec(world_up))

    def SetProj(self, angle, aspect, z_near, z_far):
        _lib().rt_camera_set_proj(C.byref(self.c), angle, aspect, z_near, z_far)

    def SetAngle(self, angle): self.SetProj(angle, self.c.aspect, self.c.z_near, self.c.z_far)
    def SetAspect(self, aspect): self.SetProj(self.c.angle, aspect, self.c.z_near, self.c.z_far)
    def SetZNear(self, z_near): self.SetProj(self.c.angle, self.c.aspect, z_near, self.c.z_far)
    def SetZFar(self, z_far): self.SetProj(self.c.angle, self.c.aspect, self.c.z_near, z_far)
    def GetEye(self): return np.array(self.c.eye, dtype=f32)
    def GetAt(self): return np.array(self.c.at, dtype=f32)
    def GetWorldUp(self): return np.array(self.c.world_up, dtype=f32)
    def GetViewMatrix(self): return _mat(self.c.view)
    def GetProj(self): return _mat(self.c.proj)
    def GetAngle(self): return self.c.angle
    def GetAspect(self): return self.c.aspect
    def GetZNear(self): return self.c.z_near
    def GetZFar(self): return self.c.z_far

    def GetViewProj(self):
        out = (C.c_float * 16)()
        _lib().rt_camera_view_proj(C.byref(self.c), out)
        return _mat(out)

    def inv_view_proj(self):
        """glm::inverse(GetViewProj()) as uploaded at App.cpp:120."""
        out = (C.c_float * 16)()
        _lib().rt_camera_inv_view_proj(C.byref(self.c), out)
        return _mat(out)


class Portal:
    """Portal.hpp / Portal.cpp:3-58."""

    def __init__(self, position, direction):
        self.m_position = np.asarray(position, dtype=f32)
        self.m_direction = np.asarray(direction, dtype=f32)

    def GetPosition(self): return self.m_position
    def GetDirection(self): return self.m_direction

    def GetDifferenceMatrixTo(self, destination_portal: "Portal"):
        out = (C.c_float * 16)()
        _lib().rt_portal_difference_matrix(_vec(self.m_direction), _vec(destination_portal.m_direction), out)
        return _mat(out)

    def RayPortal(self, ray_position, ray_direction, closest_distance, portal_width, portal_height) -> float:
        return float(_lib().rt_portal_ray(_vec(self.m_position), _vec(self.m_direction), _vec(ray_position), _vec(ray_direction), closest_distance, portal_width, portal_height))


def default_portals():
    """The two portals of App's constructor (App.cpp:31-34): directions normalised in float as there."""
    return Portal((7.0, 2.5, 1.0), normalize((2.0, 0.0, 0.5))), Portal((7.0, 1.5, 5.0), normalize((0.0, 0.0, 1.0))), 2.0, 3.0


class CameraManipulator:
    """CameraManipulator.cpp:11-210; events arrive as plain numbers instead of SDL structures."""

    def __init__(self):
        self.m = rt_manipulator()
        self.camera = None
        _lib().rt_manipulator_init(C.byref(self.m))

    def SetCamera(self, camera: Camera):
        self.camera = camera
        if camera is not None:
            _lib().rt_manipulator_set_camera(C.byref(self.m), C.byref(camera.c))

    def SetSpeed(self, speed): self.m.speed = speed
    def GetSpeed(self): return self.m.speed

    def Update(self, delta_time, portal_1: Portal, portal_2: Portal, portal_width, portal_height) -> bool:
        if self.camera is None:
            return False
        return bool(_lib().rt_manipulator_update(C.byref(self.m), C.byref(self.camera.c), delta_time, _vec(portal_1.m_position), _vec(portal_1.m_direction),
                                                 _vec(portal_2.m_position), _vec(portal_2.m_direction), portal_width, portal_height))

    def KeyboardDown(self, key, repeat=0): _lib().rt_manipulator_key(C.byref(self.m), key, 1, repeat)
    def KeyboardUp(self, key): _lib().rt_manipulator_key(C.byref(self.m), key, 0, 0)
    def MouseMove(self, xrel, yrel, left=False, right=False): _lib().rt_manipulator_mouse_move(C.byref(self.m), int(left), int(right), xrel, yrel)
    def MouseWheel(self, y): _lib().rt_manipulator_mouse_wheel(C.byref(self.m), y)


def frame_uniforms(camera: Camera, portal_1: Portal, portal_2: Portal, portal_width, portal_height, time=0.0, max_recursion_limit=5, blur_amount=0.00001) -> np.ndarray:
    """The 70-word uniform block of one draw as App::Render uploads it (App.cpp:120-138)."""
    u = api.rt_uniforms()
    L = _lib()
    L.rt_uniforms_set_camera(C.byref(u), C.byref(camera.c))
    L.rt_uniforms_set_portals(C.byref(u), _vec(portal_1.m_position), _vec(portal_1.m_direction), _vec(portal_2.m_position), _vec(portal_2.m_direction), portal_width, portal_height)
    w = np.ctypeslib.as_array(u.words).copy()
    w[19:20].view(np.uint32)[0] = np.uint32(max_recursion_limit)
    w[20] = f32(time)
    w[21] = f32(blur_amount)
    return w
