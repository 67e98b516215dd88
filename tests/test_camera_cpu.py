"""CPU tests of the camera / portal / manipulator arithmetic behind include/rt_camera.h (host code of the library).

No build of glm exists here to compare with, so bit parity with the reference's matrices is not pinned by a fixture.
What is held: bit equality with an independent step-by-step float32 restatement of the same operation order (written
below with numpy scalars), closeness to the double-precision formulas, and the behaviour of the manipulator
(/root/reference/app/core/src/CameraManipulator.cpp:27-210)."""
import math

import numpy as np
import pytest

from raytracer_amd import camera as dcam, glm_camera as g
from raytracer_amd.params import Uniforms

f32 = np.float32


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


# ---- a second restatement in numpy float32 scalars, one rounding per operation ----------------------------------
def dot3(a, b):
    return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))


def cross3(a, b):
    return np.array([f32(a[1] * b[2]) - f32(b[1] * a[2]), f32(a[2] * b[0]) - f32(b[2] * a[0]), f32(a[0] * b[1]) - f32(b[0] * a[1])], f32)


def norm3(a):
    return (a * f32(f32(1) / np.sqrt(dot3(a, a)))).astype(f32)


def look_at32(eye, at, up):
    f = norm3((at - eye).astype(f32)); s = norm3(cross3(f, up)); u = cross3(s, f)
    m = np.eye(4, dtype=f32)
    m[0][0], m[1][0], m[2][0] = s
    m[0][1], m[1][1], m[2][1] = u
    m[0][2], m[1][2], m[2][2] = -f
    m[3][0], m[3][1], m[3][2] = -dot3(s, eye), -dot3(u, eye), dot3(f, eye)
    return m


def perspective32(fovy, aspect, zn, zf):
    fovy, aspect, zn, zf = f32(fovy), f32(aspect), f32(zn), f32(zf)
    t = np.tan(f32(fovy / f32(2)), dtype=f32)
    m = np.zeros((4, 4), f32)
    m[0][0] = f32(1) / f32(aspect * t)
    m[1][1] = f32(1) / t
    m[2][2] = f32(-f32(zf + zn)) / f32(zf - zn)
    m[2][3] = -1
    m[3][2] = f32(-f32(f32(f32(2) * zf) * zn)) / f32(zf - zn)
    return m


def matmul32(a, b):
    r = np.zeros((4, 4), f32)
    for j in range(4):
        r[j] = ((a[0] * b[j][0] + a[1] * b[j][1]).astype(f32) + a[2] * b[j][2]).astype(f32) + a[3] * b[j][3]
    return r


def random_cameras(n, seed):
    rng = np.random.default_rng(seed)
    for _ in range(n):
        eye = rng.uniform(-8, 8, 3).astype(f32)
        at = (eye + rng.uniform(-3, 3, 3)).astype(f32)
        yield eye, at, np.array([0, 1, 0], f32), f32(rng.uniform(0.4, 2.4)), f32(rng.uniform(0.5, 2.5)), f32(10 ** rng.uniform(-3, 0)), f32(10 ** rng.uniform(1, 4))


def test_view_proj_bits_match_the_float_restatement():
    cam = g.Camera()
    for eye, at, up, angle, aspect, zn, zf in random_cameras(200, 5):
        cam.SetView(eye, at, up)
        cam.SetProj(angle, aspect, zn, zf)
        v, p = look_at32(eye, at, up), perspective32(angle, aspect, zn, zf)
        assert np.array_equal(bits(cam.GetViewMatrix()), bits(v))
        # the two entries that depend on tan() may differ by the last bits between numpy's and the C library's tanf
        got = cam.GetProj()
        diff = np.abs(bits(got).astype(np.int64) - bits(p).astype(np.int64))
        assert diff[0][0] <= 2 and diff[1][1] <= 2
        diff[0][0] = diff[1][1] = 0
        assert not diff.any()
        assert np.array_equal(bits(cam.GetViewProj()), bits(matmul32(got, v)))


def test_exact_known_answers():
    cam = g.Camera()
    cam.SetView((0, 0, 5), (0, 0, 0), (0, 1, 0))
    want = np.eye(4, dtype=f32); want[3][2] = -5
    assert np.array_equal(cam.GetViewMatrix(), want)
    cam.SetProj(2 * math.atan(0.5), 2.0, 1.0, 3.0)      # tan(fovy/2) = 0.5 up to rounding
    p = cam.GetProj()
    assert p[2][2] == -2 and p[3][2] == -3 and p[2][3] == -1 and p[3][3] == 0
    assert abs(p[1][1] - 2) < 1e-6 and abs(p[0][0] - 1) < 1e-6
    # inverse of the product (a permuted, scaled matrix: every cofactor is a single product)
    inv = cam.inv_view_proj().astype(np.float64).T
    vp = cam.GetViewProj().astype(np.float64).T
    assert np.allclose(inv @ vp, np.eye(4), atol=1e-6)


def test_default_camera_and_setters():
    cam = g.Camera()
    assert np.array_equal(cam.GetEye(), [1, 1, 1]) and np.array_equal(cam.GetAt(), [0, 0, 0]) and np.array_equal(cam.GetWorldUp(), [0, 1, 0])
    assert cam.GetZNear() == f32(0.01) and cam.GetZFar() == 1000 and cam.GetAspect() == 1 and cam.GetAngle() == f32(math.radians(90.0))
    before = cam.GetProj()
    cam.SetAspect(2.0)
    after = cam.GetProj()
    assert after[0][0] == f32(before[0][0] / f32(2)) or abs(after[0][0] - before[0][0] / 2) < 1e-7
    assert np.array_equal(after[1:], before[1:])
    cam.SetAngle(1.0); cam.SetZNear(0.5); cam.SetZFar(50.0)
    assert np.allclose(cam.GetProj(), perspective32(1.0, 2.0, 0.5, 50.0), rtol=3e-7, atol=0)


def test_close_to_the_double_precision_formulas():
    cam = g.Camera()
    for eye, at, up, angle, aspect, zn, zf in random_cameras(100, 6):
        cam.SetView(eye, at, up); cam.SetProj(angle, aspect, zn, zf)
        d = dcam.Camera(eye=tuple(eye), at=tuple(at), up=tuple(up), fovy=float(angle), aspect=float(aspect), z_near=float(zn), z_far=float(zf))
        assert np.allclose(cam.GetViewMatrix(), dcam.look_at(eye, at, up), rtol=1e-5, atol=1e-5)
        assert np.allclose(cam.GetProj(), dcam.perspective(float(angle), float(aspect), float(zn), float(zf)), rtol=1e-5, atol=1e-7)
        # the inverse maps clip-space points back to where the double-precision inverse does, relative to their size
        inv, dinv = cam.inv_view_proj().astype(np.float64).T, d.inv_view_proj().astype(np.float64).T
        for ndc in ((0, 0, -1, 1), (0.5, -0.25, 0, 1), (-1, 1, 0.5, 1)):
            a, b = inv @ np.array(ndc, np.float64), dinv @ np.array(ndc, np.float64)
            assert np.allclose(a[:3] / a[3], b[:3] / b[3], rtol=2e-2, atol=2e-2), (eye, at)
        # and it inverts the float matrix it was computed from
        prod = inv @ cam.GetViewProj().astype(np.float64).T
        assert np.allclose(prod, np.eye(4), atol=5e-3)


def test_portal_difference_matrix():
    p1, p2, w, h = g.default_portals()
    m12, m21 = p1.GetDifferenceMatrixTo(p2), p2.GetDifferenceMatrixTo(p1)
    d1, d2 = p1.GetDirection().astype(np.float64), p2.GetDirection().astype(np.float64)
    assert np.allclose(m12.astype(np.float64).T[:3, :3] @ d1, d2, atol=1e-6)
    assert np.allclose(m21.astype(np.float64).T[:3, :3] @ d2, d1, atol=1e-6)
    assert np.allclose(m12.astype(np.float64).T @ m21.astype(np.float64).T, np.eye(4), atol=1e-6)
    assert np.array_equal(m12[3], [0, 0, 0, 1]) and np.array_equal(m12[:, 3], [0, 0, 0, 1])
    # the double-precision helper the fixtures were made with agrees to float accuracy
    ref = dcam.Portals()
    assert np.allclose(m12, ref.m_1_to_2(), atol=1e-6) and np.allclose(m21, ref.m_2_to_1(), atol=1e-6)
    # parallel directions: identity (Portal.cpp:21-25)
    assert np.array_equal(p2.GetDifferenceMatrixTo(p2), np.eye(4, dtype=f32))
    # like the reference, no clamp of the cosine: a dot product above 1 by rounding gives NaN through acos
    a = g.Portal((0, 0, 0), np.array([0.6, 0.8, 1e-4], f32))
    b = g.Portal((0, 0, 0), np.array([0.6, 0.8, 0.0], f32))
    m = a.GetDifferenceMatrixTo(b)
    c = f32(f32(f32(0.6) * f32(0.6)) + f32(f32(0.8) * f32(0.8)))
    assert np.isnan(m).any() == bool(c > 1)


def test_ray_portal():
    p = g.Portal((7.0, 2.5, 1.0), (0.0, 0.0, 1.0))
    assert p.RayPortal((7, 2.5, -3), (0, 0, 1), 100.0, 2.0, 3.0) == 4.0
    assert p.RayPortal((7, 2.5, -3), (0, 0, 1), 3.5, 2.0, 3.0) == -1.0          # beyond the closest distance
    assert p.RayPortal((7, 2.5, 3), (0, 0, 1), 100.0, 2.0, 3.0) == -1.0           # behind the ray
    assert p.RayPortal((7, 2.5, -3), (1, 0, 0), 100.0, 2.0, 3.0) == -1.0          # parallel to the plane
    assert p.RayPortal((7.99, 3.99, -3), (0, 0, 1), 100.0, 2.0, 3.0) == 4.0       # inside the rectangle
    assert p.RayPortal((8.01, 2.5, -3), (0, 0, 1), 100.0, 2.0, 3.0) == -1.0       # outside in width
    assert p.RayPortal((7, 4.01, -3), (0, 0, 1), 100.0, 2.0, 3.0) == -1.0         # outside in height
    assert g.Portal((0, 0, 0), (0, 1, 0)).RayPortal((0, -1, 0), (0, 1, 0), 10.0, 2.0, 3.0) == -1.0   # vertical portal: no right vector


def make_rig(eye=(2, 2, 2), at=(0, 0, 0)):
    cam = g.Camera(); cam.SetView(eye, at, (0, 1, 0))
    man = g.CameraManipulator(); man.SetCamera(cam)
    return cam, man


def test_manipulator_spherical_coordinates_and_first_update():
    cam, man = make_rig()
    assert abs(man.m.distance - math.sqrt(12)) < 1e-6
    assert abs(man.m.u - math.atan2(-2, -2)) < 1e-6 and abs(man.m.v - math.acos(-2 / math.sqrt(12))) < 1e-6
    p1, p2, w, h = g.default_portals()
    assert man.Update(0.016, p1, p2, w, h) is True           # previous eye starts at the origin: the view "changed"
    assert np.allclose(cam.GetEye(), [2, 2, 2], atol=1e-5) and np.allclose(cam.GetAt(), [0, 0, 0], atol=1e-6)
    assert man.Update(0.016, p1, p2, w, h) is False          # nothing pressed, nothing moved
    assert man.Update(0.016, p1, p2, w, h) is False


def test_manipulator_keys_mouse_and_wheel():
    cam, man = make_rig(eye=(0, 0, 5), at=(0, 0, 0))
    p1, p2, w, h = g.default_portals()
    man.Update(0.0, p1, p2, w, h)
    man.KeyboardDown(g.KEY_W)
    assert man.Update(0.25, p1, p2, w, h)
    assert np.allclose(cam.GetEye(), [0, 0, 1], atol=1e-5) and np.allclose(cam.GetAt(), [0, 0, -4], atol=1e-5)   # 16 units/s forward
    man.KeyboardDown(g.KEY_S)                                  # both pressed: the later key wins, release falls back
    assert man.m.go_forward == -1
    man.KeyboardUp(g.KEY_S)
    assert man.m.go_forward == 1
    man.KeyboardUp(g.KEY_W)
    assert man.m.go_forward == 0
    man.KeyboardDown(g.KEY_SHIFT); man.KeyboardDown(g.KEY_SHIFT, repeat=1)
    assert man.GetSpeed() == 4
    man.KeyboardDown(g.KEY_D); man.KeyboardDown(g.KEY_E)
    assert man.Update(0.5, p1, p2, w, h)
    assert np.allclose(cam.GetEye(), [2, 2, 1], atol=1e-5)      # right = +x when looking down -z, up = +y
    man.KeyboardUp(g.KEY_D); man.KeyboardUp(g.KEY_E); man.KeyboardUp(g.KEY_SHIFT)
    assert man.GetSpeed() == 16 and man.m.go_right == 0 and man.m.go_up == 0
    v0, u0 = man.m.v, man.m.u
    man.MouseMove(50, 1000, left=True)
    assert man.m.u == f32(u0 + f32(0.5)) and man.m.v == f32(3.1)
    man.MouseMove(0, -100000, left=True)
    assert man.m.v == f32(0.1)
    man.MouseMove(10, 10)                                       # no button: nothing
    assert man.m.v == f32(0.1)
    d0 = man.m.distance
    man.MouseMove(0, 200, right=True)
    assert man.m.distance == f32(d0 + f32(2))
    man.MouseWheel(3)
    assert man.m.distance == f32(f32(d0 + f32(2)) + f32(f32(3 * 16) / f32(-100)))
    assert v0 != man.m.v


def test_manipulator_walks_through_a_portal():
    p1 = g.Portal((5.0, 0.0, 0.0), (0.0, 0.0, 1.0))
    p2 = g.Portal((20.0, 0.0, 10.0), (1.0, 0.0, 0.0))
    cam, man = make_rig(eye=(5.25, 0.5, 3), at=(5.25, 0.5, -2))
    man.Update(0.0, p1, p2, 2.0, 3.0)
    man.KeyboardDown(g.KEY_W)
    assert man.Update(0.25, p1, p2, 2.0, 3.0)                   # 4 units along -z: crosses portal 1 at z = 0 after 3
    m12 = p1.GetDifferenceMatrixTo(p2).astype(np.float64).T[:3, :3]
    rel = m12 @ np.array([0.25, 0.5, 0.0])                      # crossing point relative to portal 1, turned
    out_dir = m12 @ np.array([0.0, 0.0, -1.0])
    want_eye = rel + np.array([20.0, 0.0, 10.0]) + (1.0 + 0.001) * out_dir
    assert np.allclose(cam.GetEye(), want_eye, atol=1e-4)
    assert np.allclose(cam.GetAt() - cam.GetEye(), 5.0 * out_dir, atol=1e-4)   # same distance, turned view direction
    assert abs(man.m.u - math.atan2(out_dir[2], out_dir[0])) < 1e-5
    eye_after = cam.GetEye().copy()
    man.KeyboardUp(g.KEY_W)
    man.Update(0.25, p1, p2, 2.0, 3.0)            # the eye is now recomputed from (u, v): it may move by rounding once
    assert np.allclose(cam.GetEye(), eye_after, atol=1e-5)
    settled = cam.GetEye().copy()
    assert man.Update(0.25, p1, p2, 2.0, 3.0) is False
    assert np.array_equal(cam.GetEye(), settled)


def test_frame_uniforms_block():
    cam = g.Camera()
    cam.SetView((2, 2, 2), (0, 0, 0), (0, 1, 0))                # App.cpp:58-61
    cam.SetAspect(640 / 480)
    p1, p2, w, h = g.default_portals()
    u = g.frame_uniforms(cam, p1, p2, w, h, time=1.5, max_recursion_limit=5)
    d = Uniforms.from_camera(dcam.Camera(aspect=640 / 480), time=1.5, max_recursion_limit=5).pack()
    assert u.shape == d.shape == (70,)
    assert np.array_equal(u[16:25], d[16:25]) and np.array_equal(u[28:36], d[28:36]) and np.array_equal(u[68:70], d[68:70])   # positions, limits, sizes: exact
    assert np.abs(bits(u[25:28]).astype(np.int64) - bits(d[25:28]).astype(np.int64)).max() <= 1    # normalize() in float vs in double: last bit
    assert np.allclose(u[36:68], d[36:68], atol=1e-6)                                       # portal matrices
    assert np.allclose(u[0:16], d[0:16], rtol=1e-4, atol=1e-3)                               # inverse view-projection
    assert u[19:20].view(np.uint32)[0] == 5
