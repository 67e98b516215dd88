"""Scene inputs for the headless harness, tests and bench: the bundled mesh of the reference's default-style
scene and seeded synthetic triangle scenes.

The reference's default mesh (assets/xyzrgb_dragon.obj, /root/reference/app/core/src/App.cpp:17,22) is not part of
the reference checkout; like the reference's own alternative line (App.cpp:20) the bundled Stanford bunny stands in:
material 1, scale 20, Octree(18, 10, 6, 6).  Its vertex/face data is kept as a compressed array fixture
(tests/golden/bunny_mesh.npz, produced by oracle/gen_fixtures.py from assets/stanford_bunny.obj).
"""
from __future__ import annotations

import math
import os

import numpy as np

from . import api

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUNNY_FIXTURE = os.path.join(_ROOT, "tests", "golden", "bunny_mesh.npz")


def bunny_mesh(material_id: int = 1, scale: float = 20.0) -> api.Mesh:
    z = np.load(BUNNY_FIXTURE)
    return api.Mesh(positions=z["positions"], faces=z["faces"], material_id=material_id, transform=api.scale_matrix(scale))


def default_octree(device: int | None = None) -> api.Octree:
    """Bunny stand-in of the reference's scene with the reference's octree parameters (App.cpp:22).
    device: GPU that runs the construction (None: the host builder); the arrays are the same either way."""
    return api.Octree([bunny_mesh()], 18, 10, 6, 6, device=device)


# ---------------------------------------------------------------------------------------------
# procedural skybox (the reference's skybox PNGs are not redistributed; any RGBA8 cubemap is valid input)
# ---------------------------------------------------------------------------------------------
def procedural_skybox(size: int = 900, seed: int = 7) -> np.ndarray:
    """uint8 [6, size, size, 4]: sky gradient + sun disc + low-frequency clouds; deterministic."""
    rng = np.random.default_rng(seed)
    faces = np.zeros((6, size, size, 4), np.uint8)
    a = (np.arange(size, dtype=np.float64) + 0.5) / size * 2.0 - 1.0
    sc, tc = np.meshgrid(a, a)   # tc varies along rows
    one = np.ones_like(sc)
    # inverse of the face mapping used by cubemap lookup: (+x: sc=-z,tc=-y) (-x: z,-y) (+y: x,z) (-y: x,-z) (+z: x,-y) (-z: -x,-y)
    dirs = [(one, -tc, -sc), (-one, -tc, sc), (sc, one, tc), (sc, -one, -tc), (sc, -tc, one), (-sc, -tc, -one)]
    sun = np.array([0.5, 0.6, 0.3]); sun = sun / np.linalg.norm(sun)
    ph = rng.uniform(0, 2 * math.pi, size=(4, 3))
    for f, (x, y, z) in enumerate(dirs):
        n = np.sqrt(x * x + y * y + z * z)
        x, y, z = x / n, y / n, z / n
        up = np.clip(y * 0.5 + 0.5, 0, 1)
        col = np.stack([0.55 + 0.25 * (1 - up), 0.65 + 0.2 * (1 - up), 0.95 - 0.15 * (1 - up)], -1)
        ground = np.stack([0.35 + 0 * up, 0.3 + 0 * up, 0.25 + 0 * up], -1)
        col = np.where((y < -0.02)[..., None], ground, col)
        cloud = sum(np.sin(3.1 * (k + 1) * x + ph[k, 0]) * np.sin(2.7 * (k + 1) * z + ph[k, 1]) * np.cos(1.9 * (k + 1) * y + ph[k, 2]) for k in range(4)) / 4
        col = col + 0.15 * np.clip(cloud, 0, 1)[..., None] * (y > 0)[..., None]
        d = x * sun[0] + y * sun[1] + z * sun[2]
        col = col + np.clip((d - 0.995) / 0.005, 0, 1)[..., None] * np.array([1.0, 0.9, 0.6])
        faces[f, ..., :3] = np.clip(col * 255.0 + 0.5, 0, 255).astype(np.uint8)
        faces[f, ..., 3] = 255
    return faces


# ---------------------------------------------------------------------------------------------
# synthetic triangle scenes (seeded)
# ---------------------------------------------------------------------------------------------
def _icosphere(subdiv: int):
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
                  [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], np.int64)
    for _ in range(subdiv):
        cache, verts, nf = {}, list(v), []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = (verts[a] + verts[b]) / 2.0
                verts.append(m / np.linalg.norm(m))
                cache[key] = len(verts) - 1
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        v, f = np.array(verts), np.array(nf, np.int64)
    return v, f


_BOX_V = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float64)
_BOX_F = np.array([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]], np.int64)


def synthetic_primitives(n_prims: int, *, seed: int = 1, sphere_subdiv: int = 0, extent: float = 4.0, centre=(0.0, 3.0, 0.0),
                         size_range=(0.02, 0.06), materials=(1, 5, 6), perturb: float = 0.0, boxes: bool = True):
    """`n_prims` primitives as triangle meshes (alternating icospheres and boxes), uniform-random centres in a cube of
    half-size `extent` around `centre`; materials cycle through `materials`.  Returns a list of Mesh (one per material)."""
    rng = np.random.default_rng(seed)
    sv, sf = _icosphere(sphere_subdiv)
    centres = rng.uniform(-extent, extent, size=(n_prims, 3)) + np.asarray(centre)
    sizes = rng.uniform(size_range[0], size_range[1], size=n_prims)
    out_meshes = []
    per_mat_v = {m: [] for m in materials}
    per_mat_f = {m: [] for m in materials}
    per_mat_n = {m: 0 for m in materials}
    for i in range(n_prims):
        m = materials[i % len(materials)]
        if i % 2 == 0 or not boxes:
            v, f = sv, sf
        else:
            v, f = _BOX_V, _BOX_F
        vv = v * sizes[i]
        if perturb > 0.0:
            vv = vv * (1.0 + perturb * rng.standard_normal((len(v), 1)))
        per_mat_v[m].append(vv + centres[i])
        per_mat_f[m].append(f + per_mat_n[m])
        per_mat_n[m] += len(v)
    for m in materials:
        if per_mat_v[m]:
            out_meshes.append(api.Mesh(positions=np.concatenate(per_mat_v[m]).astype(np.float32), faces=np.concatenate(per_mat_f[m]).astype(np.uint32), material_id=m))
    return out_meshes


def synthetic_100k(device: int | None = None) -> api.Octree:
    """100 000 primitives (icospheres of 20 triangles and boxes of 12), seed 1, depth-8 octree: 1.6 M triangles."""
    return api.Octree(synthetic_primitives(100_000, seed=1), 8, 10, 6, 6, device=device)


def synthetic_1m(device: int | None = None) -> api.Octree:
    """1 000 perturbed icospheres of 1 280 triangles each (seed 2), depth-18 octree: 1.28 M triangles."""
    meshes = synthetic_primitives(1000, seed=2, sphere_subdiv=3, extent=3.0, size_range=(0.15, 0.4), materials=(1, 5, 6, 2), perturb=0.01, boxes=False)
    return api.Octree(meshes, 18, 10, 6, 6, device=device)


def small_test_scene(seed: int = 3, n: int = 60) -> api.Octree:
    """A few dozen primitives near the default camera's view: fast to build, exercises every material."""
    meshes = synthetic_primitives(n, seed=seed, sphere_subdiv=1, extent=1.5, centre=(0.0, 1.5, 0.0), size_range=(0.1, 0.35), materials=(0, 1, 2, 3, 4, 5, 6))
    return api.Octree(meshes, 10, 10, 6, 6)


def deep_test_meshes(seed: int = 5, n: int = 1500, far: float = 3.0e5, centre=(0.0, 3.6, 0.0)):
    """A cluster of `n` small triangles around `centre` (above the shader's big middle sphere) plus two tiny triangles
    `far` units out, which stretch the root box so much that the cluster's leaves lie about 26 levels down.  Materials 1,
    5 and 2 (metal, glass, diffuse)."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-0.25, 0.25, size=(n, 3)) + np.asarray(centre, np.float64)
    tri = c[:, None, :] + rng.normal(0.0, 0.012, size=(n, 3, 3))
    meshes = []
    for k, m in enumerate((1, 5, 2)):
        pos = tri[k::3].reshape(-1, 3)
        if k == 0:
            f = float(far)
            pos = np.concatenate([pos, [[-f, -f, -f], [-f + 1, -f, -f], [-f, -f + 1, -f], [f, f, f], [f - 1, f, f], [f, f - 1, f]]])
        meshes.append(api.Mesh(positions=pos.astype(np.float32), faces=np.arange(len(pos), dtype=np.uint32).reshape(-1, 3), material_id=m))
    return meshes


def deep_test_scene(depth_limit: int = 30, device: int | None = None) -> api.Octree:
    """A tree deeper than the reference's default limit of 18 (deep_test_meshes), for the traversal kernels built for
    deep trees."""
    return api.Octree(deep_test_meshes(), depth_limit, 10, 6, 6, device=device)


def plane_aligned_meshes(seed: int = 17):
    """Meshes whose vertices, edges and corners sit EXACTLY on the cell planes of the octree that is built from them: three
    sheets of slanted triangles over lattices with a spacing of 0.5 (every sheet vertex is a lattice point with coordinates that
    are multiples of 0.25, so every sheet edge lies in a cell plane of two axes and every vertex in planes of all three), two
    corner triangles that pin the root box to [-4, 4] x [0, 8] x [-4, 4] (cell planes at multiples of 8 / 2^depth: all exact in
    binary32), and a seeded clutter of small triangles in general position for other hits nearby.  No triangle is parallel to a
    coordinate plane.  The adversarial input for traversal orders other than the shader's: hits on box faces, edges and corners,
    where the float triangle/box test of the build and the ray's slab tests are least likely to agree."""
    def sheet(axis: int, origin, n_i: int, n_j: int, base: float):
        # lattice (i, j) over the two other axes, the sheet's own coordinate = base + 0.25 * ((i + 2 j) % 4): the three corners of
        # every triangle differ in it, so no triangle lies in a coordinate plane
        a, b = [k for k in range(3) if k != axis]
        pos = np.zeros((n_i + 1, n_j + 1, 3), np.float64)
        ii, jj = np.meshgrid(np.arange(n_i + 1), np.arange(n_j + 1), indexing="ij")
        pos[..., a] = origin[0] + 0.5 * ii
        pos[..., b] = origin[1] + 0.5 * jj
        pos[..., axis] = base + 0.25 * ((ii + 2 * jj) % 4)
        idx = lambda i, j: i * (n_j + 1) + j
        faces = []
        for i in range(n_i):
            for j in range(n_j):
                faces += [[idx(i, j), idx(i + 1, j), idx(i, j + 1)], [idx(i + 1, j), idx(i + 1, j + 1), idx(i, j + 1)]]
        return pos.reshape(-1, 3), np.array(faces, np.uint32)
    rng = np.random.default_rng(seed)
    meshes = []
    for axis, origin, n_i, n_j, base, mat in ((1, (-3.0, -3.0), 12, 12, 3.0, 1), (0, (1.0, -3.0), 10, 12, 1.0, 5), (2, (-3.0, 1.0), 12, 10, -1.0, 2)):
        p, f = sheet(axis, origin, n_i, n_j, base)
        meshes.append(api.Mesh(positions=p.astype(np.float32), faces=f, material_id=mat))
    corners = np.array([[-4.0, 0.0, -4.0], [-3.7, 0.3, -3.9], [-3.8, 0.1, -3.6], [4.0, 8.0, 4.0], [3.6, 7.9, 3.8], [3.9, 7.7, 3.7]], np.float32)
    meshes.append(api.Mesh(positions=corners, faces=np.array([[0, 1, 2], [3, 4, 5]], np.uint32), material_id=6))
    c = rng.uniform(-3.0, 3.0, size=(240, 3)) + np.array([0.0, 4.0, 0.0])
    tri = c[:, None, :] + rng.normal(0.0, 0.18, size=(240, 3, 3))
    meshes.append(api.Mesh(positions=tri.reshape(-1, 3).astype(np.float32), faces=np.arange(720, dtype=np.uint32).reshape(-1, 3), material_id=4))
    return meshes


def plane_aligned_scene(device: int | None = None) -> api.Octree:
    """plane_aligned_meshes in a depth-9 octree with small leaves (at most 4 triangles), so that the lattice planes are cell faces."""
    return api.Octree(plane_aligned_meshes(), 9, 10, 4, 6, device=device)
