"""Randomised run of the three octree builders, by hand on a GPU box (not collected by pytest):
    python tests/fuzz_octree.py [--cases N] [--seed S]

Random triangle soups (sizes from a handful to a few thousand triangles; blobs, thin slabs, axis-aligned sheets on cell
boundaries, duplicated and degenerate triangles, one or several meshes) and random builder parameters (depth limit 0..10,
node / leaf capacities and keep threshold 0..12), built by the host builder (rt_octree_build), the device builder
(rt_octree_build_device) and the octree oracle: the four arrays and the bounds must be identical word for word."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def soup(rng):
    n = int(rng.choice([1, 2, 5, 40, 300, 1500, 4000]))
    kind = int(rng.integers(0, 5))
    centres = rng.uniform(-1, 1, (n, 1, 3))
    if kind == 0:
        tri = centres + rng.normal(0, float(rng.choice([0.01, 0.05, 0.3])), (n, 3, 3))
    elif kind == 1:      # thin slabs through the middle: many triangles overlapping many children
        tri = rng.uniform(-1, 1, (n, 3, 3)) * np.array([1.0, 1.0, 0.03])
    elif kind == 2:      # axis-aligned sheets whose coordinates sit exactly on cell boundaries of a power-of-two grid
        tri = np.round((centres + rng.normal(0, 0.2, (n, 3, 3))) * 8) / 8
        tri[..., int(rng.integers(0, 3))] = np.round(centres[..., 0] * 4) / 4
    elif kind == 3:      # duplicates and degenerate triangles
        tri = centres + rng.normal(0, 0.1, (n, 3, 3))
        tri[::3] = tri[0]; tri[1::7, 2] = tri[1::7, 1]; tri[2::11] *= 1e-22
    else:                # one huge triangle among small ones
        tri = centres * 0.2 + rng.normal(0, 0.02, (n, 3, 3))
        tri[0] = np.array([[-50, -40, 0.01], [60, -45, -0.02], [3, 70, 0.0]])
    return tri.reshape(-1, 3).astype(np.float32), np.arange(3 * n, dtype=np.uint32).reshape(n, 3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    from oracle import oracle
    from raytracer_amd import api
    rng = np.random.default_rng(a.seed)
    bad = 0
    for case in range(a.cases):
        meshes, merged_v, merged_t = [], [], []
        for m in range(int(rng.choice([1, 1, 1, 2, 3]))):
            pos, faces = soup(rng)
            T = np.eye(4, dtype=np.float32)
            if rng.uniform() < 0.3:
                T = api.scale_matrix(float(rng.choice([0.5, 2.0, 20.0])))
            meshes.append(api.Mesh(positions=pos, faces=faces, material_id=int(rng.integers(0, 7)), transform=T))
        params = (int(rng.integers(0, 11)), int(rng.integers(0, 13)), int(rng.integers(0, 13)), int(rng.integers(0, 13)))
        host = api.Octree(meshes, *params).as_dict()
        dev = api.Octree(meshes, *params, device=0).as_dict()
        # the oracle builds from the merged vertex / triangle arrays (Octree.cpp:206-231: indices offset by the vertices before)
        off, vs, ns, ts = 0, [], [], []
        for me in meshes:
            t = me.m_triangles.copy(); t[:, :3] += off
            vs.append(me.m_vertecies); ts.append(t); off += len(me.m_vertecies)
        ora = oracle.octree_build(np.concatenate(vs), np.concatenate(ts), *params)
        diffs = []
        for k in ("tri_refs", "nodes", "bounds_min", "bounds_max"):
            for name, other in (("device", dev), ("oracle", ora)):
                x, y = np.ascontiguousarray(host[k]), np.ascontiguousarray(other[k])
                if x.shape != y.shape or not np.array_equal(x.view(np.uint32), y.view(np.uint32)):
                    diffs.append(f"{name}:{k}")
        for k in ("vertices", "normals"):
            x, y = np.ascontiguousarray(host[k]), np.ascontiguousarray(dev[k])
            if x.shape != y.shape or not np.array_equal(x.view(np.uint32), y.view(np.uint32)):
                diffs.append(f"device:{k}")
        if diffs:
            bad += 1
            print(f"MISMATCH case {case} seed {a.seed}: params {params}, {[len(m.m_triangles) for m in meshes]} triangles: {diffs}", flush=True)
        if case % 20 == 19:
            print(f"{case + 1} cases, {bad} mismatching", flush=True)
    print(f"done: {a.cases} cases, {bad} mismatching")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
