"""Counts, on the CPU, what a traversal over single-owner triangles with tight node boxes would visit compared with the octree
walked nearest first.  python tools/experiments/bvh_sim.py [s1m|s100k|bunny]"""
import ctypes as C, os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raytracer_amd import scenes
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "_bvh_sim.so")
subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(here, "bvh_sim.c"), "-lm"])
L = C.CDLL(so)
name = sys.argv[1] if len(sys.argv) > 1 else "s1m"
oc = {"s1m": scenes.synthetic_1m, "s100k": scenes.synthetic_100k, "bunny": scenes.default_octree}[name]()
d = oc.as_dict()
rng = np.random.default_rng(1)
lo, hi = d["bounds_min"], d["bounds_max"]; c = (lo + hi) / 2; ext = (hi - lo).max()
n = 20000
def unit(v): return v / np.linalg.norm(v, axis=1, keepdims=True)
# primary-like: from a point outside, towards points inside; secondary-like: from points inside, any direction
eye = c + np.array([0.0, 0.3, 1.6]) * ext
prim = np.concatenate([np.broadcast_to(eye, (n, 3)), unit(rng.uniform(lo, hi, (n, 3)) - eye)], 1)
sec = np.concatenate([rng.uniform(lo, hi, (n, 3)), unit(rng.normal(size=(n, 3)))], 1)
for label, rays in (("primary-like", prim), ("secondary-like", sec)):
    rays = np.ascontiguousarray(rays, np.float32)
    out = np.zeros(16, np.uint64)
    v = np.ascontiguousarray(d["vertices"], np.float32); r = np.ascontiguousarray(d["tri_refs"], np.uint32); w = np.ascontiguousarray(d["nodes"], np.uint32)
    L.bvh_sim(v.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p), C.c_uint32(len(r)), w.ctypes.data_as(C.c_void_p),
              np.ascontiguousarray(lo, np.float32).ctypes.data_as(C.c_void_p), np.ascontiguousarray(hi, np.float32).ctypes.data_as(C.c_void_p),
              rays.ctypes.data_as(C.c_void_p), C.c_uint32(n), out.ctypes.data_as(C.c_void_p))
    o = out.astype(np.float64) / n
    print(f"{name} {label}: refs {len(r)} unique {int(out[10])} nodes {int(out[11])} non-empty {int(out[12])} hit fraction {o[13]:.2f} mismatching t {int(out[9])}")
    for i, m in enumerate(("octree near-first", "tight boxes, octant order", "tight boxes, by entry distance")):
        print(f"   {m:34s} visits {o[3*i]:7.1f}  triangle tests {o[3*i+1]:7.1f}  box tests {o[3*i+2]:7.1f}")
