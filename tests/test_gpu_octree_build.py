"""The device octree builder (rt_octree_build_device) against the host builder and the reference's digests: the
flattened arrays must be identical word for word."""
import ctypes
import hashlib
import json
import os

import numpy as np
import pytest

from raytracer_amd import api, scenes

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def same(a: api.Octree, b: api.Octree):
    for k, x in a.as_dict().items():
        y = b.as_dict()[k]
        assert x.shape == y.shape, k
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), k
    assert a.max_depth == b.max_depth


def test_bunny_matches_the_survey_digests():
    oc = api.Octree([scenes.bunny_mesh()], 18, 10, 6, 6, device=0)
    want = json.load(open(os.path.join(GOLDEN, "bunny_octree_digest.json")))
    d = oc.as_dict()
    assert len(d["tri_refs"]) == 529434 and len(d["nodes"]) == 654908
    got = {k: hashlib.sha256(np.ascontiguousarray(d[k]).tobytes()).hexdigest() for k in ("vertices", "normals", "tri_refs", "nodes")}
    for k, h in got.items():
        assert want[k]["sha256"] == h, k


@pytest.mark.parametrize("params", [(18, 10, 6, 6), (6, 3, 2, 4), (10, 0, 1, 1), (8, 4, 0, 0), (12, 2, 8, 9), (1, 10, 6, 6), (0, 10, 6, 6), (5, 1, 3, 8)])
@pytest.mark.parametrize("seed", [1, 2])
def test_equals_host_builder_on_random_soups(params, seed):
    rng = np.random.default_rng(seed)
    n = 3000
    centres = rng.uniform(-1, 1, (n, 1, 3))
    tri = centres + rng.normal(0, [0.02, 0.3][seed - 1], (n, 3, 3))
    tri[::50] *= 1e-20            # a cluster of (near-)degenerate triangles at the origin, denormal edge lengths
    pos = tri.reshape(-1, 3).astype(np.float32)
    faces = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    meshes = [api.Mesh(positions=pos, faces=faces, material_id=seed)]
    same(api.Octree(meshes, *params), api.Octree(meshes, *params, device=0))


def test_small_scene_and_multi_mesh():
    meshes = scenes.synthetic_primitives(200, seed=5, sphere_subdiv=1, extent=1.5, centre=(0.0, 1.5, 0.0), size_range=(0.1, 0.35), materials=(0, 1, 2, 3, 4, 5, 6))
    same(api.Octree(meshes, 10, 10, 6, 6), api.Octree(meshes, 10, 10, 6, 6, device=0))


def test_empty_and_single_triangle():
    empty = api.Mesh(positions=np.zeros((0, 3), np.float32), faces=np.zeros((0, 3), np.uint32))
    same(api.Octree([empty]), api.Octree([empty], device=0))
    one = api.Mesh(positions=np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), faces=np.array([[0, 1, 2]], np.uint32))
    same(api.Octree([one], 18, 10, 0, 6), api.Octree([one], 18, 10, 0, 6, device=0))


def test_large_synthetic_scene_equals_host_builder():
    meshes = scenes.synthetic_primitives(20_000, seed=1)
    same(api.Octree(meshes, 8, 10, 6, 6), api.Octree(meshes, 8, 10, 6, 6, device=0))


def test_invalid_device_is_rejected():
    with pytest.raises(api.RtError):
        api.Octree([scenes.bunny_mesh()], device=99)


@pytest.mark.parametrize("params", [(8, 10, 6, 6), (7, 3, 2, 8), (6, 0, 4, 1)])
def test_many_keep_candidates_with_equal_overlap_counts(params):
    """Hundreds of large triangles through the centre of the box overlap all eight children of the root and most children
    further down: far more "keep" candidates than a node may hold, nearly all with the same overlap count, so the order
    std::partial_sort leaves among equals decides which stay and in which order the rest are pushed down."""
    rng = np.random.default_rng(12)
    n_big, n_small = 400, 4000
    big = rng.uniform(-1.0, 1.0, (n_big, 3, 3)) * np.array([1.0, 1.0, 0.05])          # thin slabs through the middle
    big[: n_big // 4] = big[0]                                                          # a hundred identical triangles
    small = rng.uniform(-1, 1, (n_small, 1, 3)) + rng.normal(0, 0.03, (n_small, 3, 3))
    tri = np.concatenate([big, small])
    order = rng.permutation(len(tri))
    pos = tri[order].reshape(-1, 3).astype(np.float32)
    faces = np.arange(3 * len(tri), dtype=np.uint32).reshape(-1, 3)
    meshes = [api.Mesh(positions=pos, faces=faces, material_id=3)]
    host, dev = api.Octree(meshes, *params), api.Octree(meshes, *params, device=0)
    same(host, dev)
    assert len(host.m_compressed_triangles) > 3 * len(tri)      # the slabs are referenced from many nodes


def test_flat_mesh_and_huge_triangle():
    """A mesh of zero thickness (its box is flat: strict comparisons in the overlap test matter) plus one triangle that
    dwarfs the rest."""
    g = np.linspace(-1, 1, 40, dtype=np.float32)
    xx, yy = np.meshgrid(g, g)
    pos = np.stack([xx.ravel(), yy.ravel(), np.zeros(xx.size, np.float32)], 1)
    idx = np.arange(40 * 40).reshape(40, 40)
    faces = np.concatenate([np.stack([idx[:-1, :-1], idx[:-1, 1:], idx[1:, :-1]], -1).reshape(-1, 3), np.stack([idx[1:, :-1], idx[:-1, 1:], idx[1:, 1:]], -1).reshape(-1, 3)]).astype(np.uint32)
    flat = api.Mesh(positions=pos, faces=faces, material_id=1)
    huge = api.Mesh(positions=np.array([[-50, -50, 0], [50, -50, 0], [0, 80, 0]], np.float32), faces=np.array([[0, 1, 2]], np.uint32), material_id=2)
    for meshes in ([flat], [flat, huge]):
        same(api.Octree(meshes, 12, 10, 6, 6), api.Octree(meshes, 12, 10, 6, 6, device=0))


def test_scene_straight_from_meshes_renders_what_the_two_steps_render():
    """rt_scene_create_from_meshes (build on the device, scene from the arrays where they are) against Octree + Scene: same
    bounds, same records, same pixels and the same work counted, in all three pipelines."""
    from raytracer_amd.camera import Camera
    from raytracer_amd.params import Uniforms
    meshes = [scenes.bunny_mesh()] + scenes.synthetic_primitives(40, seed=9, sphere_subdiv=1, extent=1.2, centre=(0.0, 1.5, 0.0), size_range=(0.1, 0.3), materials=(0, 2, 5))
    oc = api.Octree(meshes, 12, 10, 6, 6)
    sky = np.random.default_rng(3).integers(0, 256, size=(6, 16, 16, 4), dtype=np.uint8)
    two = api.Scene.from_octree(oc, device=0); two.set_skybox(sky)
    one = api.Scene.from_meshes(meshes, 12, 10, 6, 6, device=0); one.set_skybox(sky)
    assert np.array_equal(one.bounds_min.view(np.uint32), oc.GetMinBounds().view(np.uint32))
    assert np.array_equal(one.bounds_max.view(np.uint32), oc.GetMaxBounds().view(np.uint32))
    def info(sc):      # node records, deepest level, references, face-coplanar triangles
        out = (ctypes.c_uint32 * 4)()
        assert api.lib().rt_internal_scene_info(sc._h, out) == api.RT_OK
        return tuple(out)
    assert info(one) == info(two)
    u = Uniforms.from_camera(Camera(aspect=160 / 96), max_recursion_limit=4).pack()
    for pipe in ("megakernel", "wavefront", "persistent"):
        outs = []
        for sc in (one, two):
            sc.set_pipeline(pipe); sc.set_counting(True)
            g = sc.render(u, 160, 96)
            outs.append((g, sc.stats()))
        (a, sa), (b, sb) = outs
        for k in ("kind", "slot", "nrays"):
            assert np.array_equal(a[k], b[k]), (pipe, k)
        for k in ("t", "depth", "rgba"):
            assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), (pipe, k)
        assert sa == sb, pipe
    one.close(); two.close()


def test_scene_from_meshes_refuses_what_scene_create_refuses():
    bad = scenes.bunny_mesh()
    bad.m_triangles = bad.m_triangles.copy()
    bad.m_triangles[5, 3] = 9          # no such material
    with pytest.raises(api.RtError):
        api.Scene.from_meshes([bad], device=0)
    empty = api.Scene.from_meshes([], device=0)      # no mesh at all: a scene of spheres, like rt_scene_create with empty arrays
    empty.close()
