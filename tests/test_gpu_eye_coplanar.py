"""The camera position IN the plane of scene triangles (exact near-first order): a primary ray that also lies in that plane gives the
triangle test a determinant of rounding noise, and what the shader reports then depends on its visiting order alone -- so the library
lists the frame's eye-coplanar triangles and sends the rays concerned through the shader's order (the llvmpipe record of such a pixel
is tests/golden/refscene_planes_coplanar.npz, compared in test_gpu_golden.py).  Here: a slanted sheet whose plane x + y = 4 contains
the eye exactly, small enough to be listed triangle by triangle and large enough to overflow the list (every primary ray of the frame
then takes the shader's order), against the CPU restatement, bit for bit, with the work counters telling which path ran."""
import numpy as np
import pytest

from raytracer_amd import api
from raytracer_amd.camera import Camera
from raytracer_amd.params import Uniforms

pytestmark = pytest.mark.gpu


def sheet_scene(n_x: int, n_z: int):
    """A sheet of 2 n_x n_z triangles in the plane x + y = 4 over a lattice of spacing 0.5 (all coordinates exact in binary32), two
    triangles that pin the root box, and seeded clutter behind the sheet."""
    ii, kk = np.meshgrid(np.arange(n_x + 1), np.arange(n_z + 1), indexing="ij")
    pos = np.stack([0.5 * ii, 4.0 - 0.5 * ii, -0.25 * n_z + 0.5 * kk], -1).reshape(-1, 3).astype(np.float32)
    idx = lambda i, k: i * (n_z + 1) + k
    faces = []
    for i in range(n_x):
        for k in range(n_z):
            faces += [[idx(i, k), idx(i + 1, k), idx(i, k + 1)], [idx(i + 1, k), idx(i + 1, k + 1), idx(i, k + 1)]]
    meshes = [api.Mesh(positions=pos, faces=np.array(faces, np.uint32), material_id=1)]
    corners = np.array([[-4.0, 0.0, -4.0], [-3.7, 0.3, -3.9], [-3.8, 0.1, -3.6], [4.0, 8.0, 4.0], [3.6, 7.9, 3.8], [3.9, 7.7, 3.7]], np.float32)
    meshes.append(api.Mesh(positions=corners, faces=np.array([[0, 1, 2], [3, 4, 5]], np.uint32), material_id=6))
    rng = np.random.default_rng(3)
    c = rng.uniform(-1.0, 3.5, size=(200, 3)) + np.array([0.5, 0.5, -1.0])
    tri = c[:, None, :] + rng.normal(0.0, 0.2, size=(200, 3, 3))
    meshes.append(api.Mesh(positions=tri.reshape(-1, 3).astype(np.float32), faces=np.arange(600, dtype=np.uint32).reshape(-1, 3), material_id=4))
    return api.Octree(meshes, 9, 10, 4, 6, device=0)


@pytest.mark.parametrize("n_x,n_z,overflow", [(2, 2, False), (6, 8, True)])
def test_eye_in_the_plane_of_a_sheet(test_sky, n_x, n_z, overflow):
    from oracle import oracle
    oc = sheet_scene(n_x, n_z)
    osc = oracle.OracleScene(oc.as_dict(), test_sky)
    sc = api.Scene.from_octree(oc, device=0)
    sc.set_skybox(test_sky)
    w, h = 129, 65          # odd: the centre column of rays has ndc x = 0
    # eye and target in the sheet's plane (x + y = 4), the view's up vector in it as well: a whole column of primary rays lies in the plane
    cam = Camera(eye=(-2.0, 6.0, 0.25), at=(1.0, 3.0, 0.25), up=(0.0, 0.0, 1.0), fovy=0.7, aspect=w / h)
    for limit in (1, 4):
        u = Uniforms.from_camera(cam, max_recursion_limit=limit, blur_amount=0.0).pack()
        want = osc.render(u, w, h)
        assert int((want["kind"] == api.HIT_TRIANGLE).sum()) > 500
        for pipe in ("wavefront", "persistent"):
            sc.set_pipeline(pipe)
            sc.set_traversal_order("near_first_exact")
            got = sc.render(u, w, h)
            assert sc.last_mode() == (pipe, "near_first_exact")
            for k in ("kind", "slot", "nrays"):
                assert np.array_equal(got[k], want[k]), (pipe, limit, k)
            for k in ("rgba", "depth", "t"):
                assert np.array_equal(np.ascontiguousarray(got[k]).view(np.uint32), np.ascontiguousarray(want[k]).view(np.uint32)), (pipe, limit, k)
            if limit == 1:
                # only first segments: with the list overflowing every one of them starts in the shader's order, otherwise only the
                # few whose direction lies in the sheet's plane and whose own test accepts a listed triangle
                sc.set_counting(True)
                sc.render(u, w, h)
                st = sc.stats()
                sc.set_counting(False)
                assert st["rays_entering_octree"] > 1000
                if overflow:
                    assert st["exact_fallbacks"] >= st["rays_entering_octree"]
                else:
                    assert st["exact_fallbacks"] < st["rays_entering_octree"] // 4
    sc.close()
