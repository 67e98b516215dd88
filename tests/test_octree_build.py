"""CPU tests of the host-side scene build (rt_mesh_* / rt_octree_* in librt_hotpath.so) against the octree oracle,
the digests of the reference build's arrays, and the node-word format the shader reads."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import oracle
from raytracer_amd import api, scenes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_bunny_matches_the_survey_digests(bunny_octree):
    """bunny_octree_digest.json holds the digests that a build of the reference's Octree.cpp against a stand-in for glm produced when
    the reference was surveyed: a cross-check of the builders, not a pin by a reference build (glm is absent from the image)."""
    d = json.load(open(os.path.join(GOLDEN, "bunny_octree_digest.json")))
    o = bunny_octree
    for key, arr in (("vertices", o.m_vertecies), ("normals", o.m_normals), ("tri_refs", o.m_compressed_triangles), ("nodes", o.m_compressed_node_buffer)):
        assert arr.nbytes == d[key]["bytes"], key
        assert hashlib.sha256(arr.tobytes()).hexdigest() == d[key]["sha256"], key
    b = np.concatenate([o.GetMinBounds(), o.GetMaxBounds()]).astype(np.float32)
    assert hashlib.sha256(b.tobytes()).hexdigest() == d["bounds"]["sha256"]
    assert (len(o.m_vertecies), len(o.m_compressed_triangles), len(o.m_compressed_node_buffer), o.max_depth) == (35947, 529434, 654908, 17)


def test_oracle_builder_matches_the_survey_digests():
    z = np.load(os.path.join(GOLDEN, "bunny_mesh.npz"))
    v, n, t = oracle.mesh_build(z["positions"], None, z["faces"], 1, np.diag([20.0, 20.0, 20.0, 1.0]).astype(np.float32))
    o = oracle.octree_build(v, t, 18, 10, 6, 6)
    d = json.load(open(os.path.join(GOLDEN, "bunny_octree_digest.json")))
    for key, arr in (("vertices", v), ("normals", n), ("tri_refs", o["tri_refs"]), ("nodes", o["nodes"])):
        assert hashlib.sha256(arr.tobytes()).hexdigest() == d[key]["sha256"], key


def random_soup(rng, n_tri, spread=1.0, size=0.2):
    c = rng.uniform(-spread, spread, size=(n_tri, 1, 3))
    p = (c + rng.uniform(-size, size, size=(n_tri, 3, 3))).reshape(-1, 3).astype(np.float32)
    f = np.arange(3 * n_tri, dtype=np.uint32).reshape(-1, 3)
    return p, f


@pytest.mark.parametrize("params", [(18, 10, 6, 6), (6, 4, 2, 3), (3, 1, 1, 9), (10, 0, 6, 1), (1, 10, 6, 6), (12, 3, 0, 2)])
@pytest.mark.parametrize("seed", [0, 1])
def test_builder_equals_oracle_on_random_soups(params, seed):
    rng = np.random.default_rng(seed)
    p, f = random_soup(rng, 3000, size=0.15 if seed else 0.6)
    m = api.Mesh(positions=p, faces=f, material_id=3, transform=api.scale_matrix(1.5))
    ov, on, ot = oracle.mesh_build(p, None, f, 3, api.scale_matrix(1.5))
    assert np.array_equal(m.m_vertecies.view(np.uint32), ov.view(np.uint32)) and np.array_equal(m.m_normals.view(np.uint32), on.view(np.uint32)) and np.array_equal(m.m_triangles, ot)
    o = api.Octree([m], *params)
    r = oracle.octree_build(ov, ot, *params)
    assert np.array_equal(o.m_compressed_node_buffer, r["nodes"])
    assert np.array_equal(o.m_compressed_triangles, r["tri_refs"])
    assert np.array_equal(o.GetMinBounds(), r["bounds_min"]) and np.array_equal(o.GetMaxBounds(), r["bounds_max"])
    assert o.max_depth == r["max_depth"]


@pytest.mark.parametrize("params", [(8, 10, 6, 6), (7, 3, 2, 8), (6, 0, 4, 1)])
def test_builder_equals_oracle_with_many_equal_keep_candidates(params):
    """Far more "keep" candidates than a node may hold, nearly all with the same overlap count (thin slabs through the centre,
    a hundred of them identical): which stay and in what order the rest go down is decided by the order std::partial_sort
    leaves among equals (Octree.cpp:172-181)."""
    rng = np.random.default_rng(12)
    big = rng.uniform(-1.0, 1.0, (400, 3, 3)) * np.array([1.0, 1.0, 0.05])
    big[:100] = big[0]
    small = rng.uniform(-1, 1, (4000, 1, 3)) + rng.normal(0, 0.03, (4000, 3, 3))
    tri = np.concatenate([big, small])
    p = tri[rng.permutation(len(tri))].reshape(-1, 3).astype(np.float32)
    f = np.arange(len(p), dtype=np.uint32).reshape(-1, 3)
    m = api.Mesh(positions=p, faces=f, material_id=3)
    ov, on, ot = oracle.mesh_build(p, None, f, 3, api.IDENTITY)
    o = api.Octree([m], *params)
    r = oracle.octree_build(ov, ot, *params)
    assert np.array_equal(o.m_compressed_node_buffer, r["nodes"]) and np.array_equal(o.m_compressed_triangles, r["tri_refs"])


def test_multi_mesh_merge_offsets_indices():
    rng = np.random.default_rng(5)
    p1, f1 = random_soup(rng, 500)
    p2, f2 = random_soup(rng, 700, spread=2.0)
    n2 = rng.standard_normal((len(p2), 3)).astype(np.float32)
    m1 = api.Mesh(positions=p1, faces=f1, material_id=1)
    m2 = api.Mesh(positions=p2, normals=n2, faces=f2, material_id=2)       # file normals are taken verbatim (Mesh.cpp:46-50)
    assert np.array_equal(m2.m_normals[:, :3], n2) and not m2.m_normals[:, 3].any()
    o = api.Octree([m1, m2], 8, 10, 6, 6)
    v = np.concatenate([m1.m_vertecies, m2.m_vertecies])
    t2 = m2.m_triangles.copy(); t2[:, :3] += len(m1.m_vertecies)
    r = oracle.octree_build(v, np.concatenate([m1.m_triangles, t2]), 8, 10, 6, 6)
    assert np.array_equal(o.m_vertecies, v) and np.array_equal(o.m_normals, np.concatenate([m1.m_normals, m2.m_normals]))
    assert np.array_equal(o.m_compressed_node_buffer, r["nodes"]) and np.array_equal(o.m_compressed_triangles, r["tri_refs"])
    assert set(np.unique(o.m_compressed_triangles[:, 3])) == {1, 2}


def walk(nodes, refs):
    """Decode the node words the way the shader does (ray_tracer.frag:460-464,481-484); returns (n_nodes, triangle slots seen, max depth)."""
    stack, seen, n_nodes, max_depth = [(0, 0)], 0, 0, 0
    covered = np.zeros(len(refs), bool)
    while stack:
        p, d = stack.pop()
        info, start = int(nodes[p]), int(nodes[p + 1])
        cc, mask, tc = info & 0xF, (info >> 8) & 0xFF, info >> 16
        assert bin(mask).count("1") == cc
        assert not covered[start:start + tc].any()
        covered[start:start + tc] = True
        n_nodes += 1; seen += tc; max_depth = max(max_depth, d)
        for k in range(cc):
            stack.append((int(nodes[p + 2 + k]), d + 1))
    assert covered.all()
    return n_nodes, seen, max_depth


def test_node_words_are_a_consistent_tree(bunny_octree):
    o = bunny_octree
    n_nodes, seen, depth = walk(o.m_compressed_node_buffer, o.m_compressed_triangles)
    assert seen == len(o.m_compressed_triangles)
    assert depth == o.max_depth == 17 and n_nodes == 218303
    # every triangle of the mesh is referenced at least once and duplicates carry identical records
    uniq = np.unique(o.m_compressed_triangles, axis=0)
    assert len(uniq) == 69451


def test_transform_drops_translation():
    """Mesh.cpp:44 multiplies vec4(x, y, z, 0): a translation column has no effect."""
    T = api.scale_matrix(2.0).copy()
    T[3, :3] = (5.0, 6.0, 7.0)   # [column][row]: column 3 is the translation
    p = np.array([[1, 2, 3], [0, 1, 0], [1, 0, 0]], np.float32)
    m = api.Mesh(positions=p, faces=np.array([[0, 1, 2]], np.uint32), material_id=0, transform=T)
    assert np.array_equal(m.m_vertecies[:, :3], p * 2.0) and not m.m_vertecies[:, 3].any()


def test_normals_accumulate_from_seed_vector():
    """Without file normals each vertex normal is normalize((1,0,0) + sum of face normals) (Mesh.cpp:52,79-83,96-99)."""
    p = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [5, 5, 5]], np.float32)
    m = api.Mesh(positions=p, faces=np.array([[0, 1, 2]], np.uint32), material_id=0)
    expect = np.array([1.0, 0.0, 1.0]) / np.sqrt(2.0)
    assert np.allclose(m.m_normals[:3, :3], expect, atol=1e-6)
    assert np.array_equal(m.m_normals[3, :3], np.array([1, 0, 0], np.float32))   # unreferenced vertex keeps the seed


def test_obj_reader(tmp_path):
    tri = tmp_path / "t.obj"
    tri.write_text("# c\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nf 1 2 3\nf 1//1 3//3 4//4\nf -4 -3 -1\n")
    m = api.Mesh(str(tri), 4)
    assert m.m_triangles.tolist() == [[0, 1, 2, 4], [0, 2, 3, 4], [0, 1, 3, 4]]
    # quads: cut along the shorter diagonal (the rule of the OBJ library the reference links, which is not part of the
    # reference checkout: PARITY UNPINNED); larger polygons are refused
    quad = tmp_path / "q.obj"
    quad.write_text("v 0 0 0\nv 1 0 0\nv 3 1 0\nv 0 1 0\n"      # |v0v2|^2 = 10 > |v1v3|^2 = 2  -> (0,1,3) (1,2,3)
                    "v 0 0 1\nv 3 0 1\nv 1 1 1\nv 0 1 1\n"      # |v0v2|^2 = 2 < |v1v3|^2 = 10  -> (0,1,2) (0,2,3)
                    "v 0 0 2\nv 1 0 2\nv 1 1 2\nv 0 1 2\n"      # equal diagonals               -> (0,1,3) (1,2,3)
                    "f 1 2 3 4\nf 5 6 7 8\nf 9//9 10//10 11//11 12//12\nf 1 2 3\n")
    q = api.Mesh(str(quad), 2)
    assert q.m_triangles[:, :3].tolist() == [[0, 1, 3], [1, 2, 3], [4, 5, 6], [4, 6, 7], [8, 9, 11], [9, 10, 11], [0, 1, 2]]
    assert np.array_equal(oracle.parse_obj(str(quad))[2], q.m_triangles[:, :3])
    penta = tmp_path / "p.obj"
    penta.write_text("v 0 0 0\nv 1 0 0\nv 2 1 0\nv 1 2 0\nv 0 1 0\nf 1 2 3 4 5\n")
    with pytest.raises(api.RtError) as e:
        api.Mesh(str(penta), 0)
    assert e.value.code == api.RT_ESCENE and "four" in str(e.value)
    with pytest.raises(api.RtError):
        api.Mesh(str(tmp_path / "missing.obj"), 0)
    bad = tmp_path / "n.obj"
    bad.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1\n")
    with pytest.raises(api.RtError):           # Mesh.cpp:28-33: vertex and normal counts differ
        api.Mesh(str(bad), 0)


def test_oracle_obj_reader_agrees_with_product(tmp_path):
    z = np.load(os.path.join(GOLDEN, "bunny_mesh.npz"))
    path = tmp_path / "b.obj"
    with open(path, "w") as f:
        for p in z["positions"][:300]:
            f.write("v %r %r %r\n" % tuple(float(x) for x in p))
        for a, b, c in z["faces"][(z["faces"] < 300).all(1)]:
            f.write(f"f {a + 1} {b + 1} {c + 1}\n")
    pos, nrm, faces = oracle.parse_obj(str(path))
    m = api.Mesh(str(path), 1, api.scale_matrix(20.0))
    v, n, t = oracle.mesh_build(pos, nrm, faces, 1, api.scale_matrix(20.0))
    assert np.array_equal(m.m_vertecies, v) and np.array_equal(m.m_normals.view(np.uint32), n.view(np.uint32)) and np.array_equal(m.m_triangles, t)


def test_tri_box_edge_cases():
    c, h = [0, 0, 0], [1, 1, 1]
    assert oracle.tri_box(c, h, [[-0.5, 0, 0], [0.5, 0, 0], [0, 0.5, 0]])            # inside
    assert not oracle.tri_box(c, h, [[2, 2, 2], [3, 2, 2], [2, 3, 2]])               # far away
    assert oracle.tri_box(c, h, [[1, -3, -3], [1, 3, -3], [1, 0, 3]])                # lies in the face plane x = 1 (touching counts)
    assert not oracle.tri_box(c, h, [[1.0001, -3, -3], [1.0001, 3, -3], [1.0001, 0, 3]])
    assert oracle.tri_box(c, h, [[-5, -5, 0], [5, -5, 0], [0, 5, 0]])                # box inside a big triangle
    assert oracle.tri_box(c, h, [[2.2, 0, 0], [0, 2.2, 0], [0, 0, 2.2]])              # plane x+y+z = 2.2 cuts the corner (1,1,1)
    assert not oracle.tri_box(c, h, [[3.1, 0, 0], [0, 3.1, 0], [0, 0, 3.1]])          # plane x+y+z = 3.1 misses it although the bounding boxes overlap


def test_depth_limit_guard():
    m = scenes.bunny_mesh()
    with pytest.raises(api.RtError) as e:
        api.Octree([m], 1000, 10, 6, 6)
    assert e.value.code == api.RT_ELIMIT
