#!/usr/bin/env python3
"""Development aid (GPU box): a scene built against the exact near-first order's SECONDARY rays -- a perfectly specular floor (the
roughness-0 dielectric) under a vertical sheet of triangles in the plane x = c, the eye in that plane looking down at the floor within
it, odd frame width: the centre column's primary rays lie in the plane, and so do their reflections and refractions, origin and
direction both, so the sheet's triangles meet secondary rays with a determinant of rounding noise.  Compares both wave-wide pipelines in
the exact order with the CPU restatement."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle
from raytracer_amd import api
from raytracer_amd.camera import Camera
from raytracer_amd.params import Uniforms

def scene(c, depth, phi=0.0):
    meshes = []
    floor = np.array([[-4, 0.25, -4], [4, 0.25, -4], [4, 0.25, 4], [-4, 0.25, 4]], np.float32)
    meshes.append(api.Mesh(positions=floor, faces=np.array([[0, 1, 2], [0, 2, 3]], np.uint32), material_id=6))
    ny, nz = 6, 10
    jj, kk = np.meshgrid(np.arange(ny + 1), np.arange(nz + 1), indexing="ij")
    # the sheet's vertical plane passes through (c, *, 0) and is turned about the y axis by phi
    along = -2.0 + 0.4 * kk
    pos = np.stack([c - np.sin(phi) * along, 0.75 + 0.4 * jj, np.cos(phi) * along], -1).reshape(-1, 3).astype(np.float32)
    idx = lambda j, k: j * (nz + 1) + k
    faces = []
    for j in range(ny):
        for k in range(nz):
            faces += [[idx(j, k), idx(j + 1, k), idx(j, k + 1)], [idx(j + 1, k), idx(j + 1, k + 1), idx(j, k + 1)]]
    meshes.append(api.Mesh(positions=pos, faces=np.array(faces, np.uint32), material_id=1))
    corners = np.array([[-4.0, 0.0, -4.0], [-3.7, 0.3, -3.9], [-3.8, 0.1, -3.6], [4.0, 8.0, 4.0], [3.6, 7.9, 3.8], [3.9, 7.7, 3.7]], np.float32)
    meshes.append(api.Mesh(positions=corners, faces=np.array([[0, 1, 2], [3, 4, 5]], np.uint32), material_id=2))
    rng = np.random.default_rng(5)
    cc = rng.uniform(-3.0, 3.0, size=(150, 3)) + np.array([0.0, 3.5, 0.0])
    tri = cc[:, None, :] + rng.normal(0.0, 0.2, size=(150, 3, 3))
    meshes.append(api.Mesh(positions=tri.reshape(-1, 3).astype(np.float32), faces=np.arange(450, dtype=np.uint32).reshape(-1, 3), material_id=4))
    return api.Octree(meshes, depth, 10, 4, 6, device=0)

bad_total = 0
sky = np.random.default_rng(1).integers(0, 256, size=(6, 32, 32, 4), dtype=np.uint8)
for c, depth, phi in ((0.53125, 6, 0.0), (0.3, 7, 0.4), (-1.1875, 5, -0.9), (0.5, 8, 0.2), (0.0, 7, 0.7853981), (0.7, 6, 1.2)):
    oc = scene(c, depth, phi)
    osc = oracle.OracleScene(oc.as_dict(), sky)
    sc = api.Scene.from_octree(oc, device=0); sc.set_skybox(sky)
    info = sc.info() if hasattr(sc, "info") else None
    for eye_y, eye_z, at_z in ((3.0, -3.5, 0.0), (5.0, 3.0, -1.0), (1.5, -3.9, 1.0), (6.0, 0.125, 0.125)):
        for t in range(6):
            w, h = 129, 97
            ex, ez = c - np.sin(phi) * eye_z, np.cos(phi) * eye_z          # eye and target in the sheet's plane
            ax, az = c - np.sin(phi) * at_z, np.cos(phi) * at_z
            cam = Camera(eye=(float(np.float32(ex)), eye_y, float(np.float32(ez))), at=(float(np.float32(ax)), 0.25, float(np.float32(az))),
                         up=(0.0, 1.0, 0.0) if abs(eye_z - at_z) > 1e-6 else (float(-np.sin(phi)), 0.0, float(np.cos(phi))), fovy=0.9, aspect=w / h)
            u = Uniforms.from_camera(cam, time=float(t) * 0.37, max_recursion_limit=5, blur_amount=0.0).pack()
            want = osc.render(u, w, h)
            for pipe in ("wavefront", "persistent"):
                sc.set_pipeline(pipe); sc.set_traversal_order("near_first_exact")
                got = sc.render(u, w, h)
                mode = sc.last_mode()
                bad = int((np.ascontiguousarray(got["rgba"]).view(np.uint32) != np.ascontiguousarray(want["rgba"]).view(np.uint32)).any(-1).sum())
                bad_total += bad
                if bad:
                    ys, xs = np.nonzero((np.ascontiguousarray(got["rgba"]).view(np.uint32) != np.ascontiguousarray(want["rgba"]).view(np.uint32)).any(-1))
                    print(f"c={c} depth={depth} phi={phi} eye=({eye_y},{eye_z}) t={t} {pipe} {mode}: {bad} differing pixels, columns {sorted(set(xs.tolist()))[:8]}", flush=True)
    sc.close()
print("total differing pixels:", bad_total)
