"""Randomised run of rt_scene_create's validation, by hand on a GPU box (not collected by pytest):
    python tests/fuzz_scene_validation.py [--cases N] [--seed S]

Corrupts a few words of a valid scene's node buffer or triangle references (random values, off-by-one counts, offsets
pointing backwards to make cycles or sideways to share subtrees, truncated buffers) and creates a scene from them.  The
library must either refuse the scene (RT_ESCENE / RT_ELIMIT with a message) or accept it and then render a small frame
without reading out of bounds; which of the two is printed in the summary.  A crash or a GPU fault is the failure."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=1000)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    from raytracer_amd import api, scenes
    from raytracer_amd.camera import Camera
    from raytracer_amd.params import Uniforms
    rng = np.random.default_rng(a.seed)
    base = scenes.small_test_scene().as_dict()
    u = Uniforms.from_camera(Camera(eye=(1.5, 2.0, 2.5), at=(0.0, 1.5, 0.0), aspect=1.0), max_recursion_limit=3).pack()
    refused, accepted, codes = 0, 0, {}
    for case in range(a.cases):
        d = {k: np.array(v, copy=True) for k, v in base.items()}
        nodes, refs = d["nodes"], d["tri_refs"]
        for _ in range(int(rng.integers(1, 6))):
            kind = int(rng.integers(0, 7))
            i = int(rng.integers(0, len(nodes)))
            if kind == 0:
                nodes[i] = rng.integers(0, 1 << 32, dtype=np.uint64).astype(np.uint32)
            elif kind == 1:
                nodes[i] = np.uint32(rng.integers(0, len(nodes) + 3))                 # a plausible offset: backwards (cycle), sideways (shared subtree), just past the end
            elif kind == 2:
                nodes[i] = np.uint32((int(nodes[i]) + int(rng.choice([-1, 1, 1 << 8, 1 << 16, -(1 << 16)]))) & 0xFFFFFFFF)
            elif kind == 3:
                nodes[i] = np.uint32(0 if rng.uniform() < 0.5 else 0xFFFFFFFF)
            elif kind == 4:
                j = int(rng.integers(0, len(refs)))
                refs[j, int(rng.integers(0, 4))] = np.uint32(rng.choice([len(d["vertices"]), len(d["vertices"]) + 7, 7, 0xFFFFFFFF, int(rng.integers(0, len(d["vertices"])))]))
            elif kind == 5:
                d["nodes"] = nodes = nodes[: int(rng.integers(0, len(nodes)))].copy() if len(nodes) > 1 else nodes
                if len(nodes) == 0:
                    break
            else:
                d["tri_refs"] = refs = refs[: int(rng.integers(1, len(refs)))].copy()
        try:
            sc = api.Scene.from_dict(d)
        except api.RtError as e:
            refused += 1
            codes[e.code] = codes.get(e.code, 0) + 1
            assert e.code in (api.RT_ESCENE, api.RT_ELIMIT), e
            assert len(str(e)) > 20
            continue
        accepted += 1
        for pipe in ("megakernel", "wavefront", "persistent"):
            sc.set_pipeline(pipe)
            r = sc.render(u, 48, 48)
            assert r["rgba"].shape == (48, 48, 4)
        sc.close()
        if case % 100 == 99:
            print(f"{case + 1} cases: {refused} refused {codes}, {accepted} accepted and rendered", flush=True)
    print(f"done: {a.cases} cases: {refused} refused {codes}, {accepted} accepted and rendered")
    return 0


if __name__ == "__main__":
    sys.exit(main())
