"""Randomised run of the sequence / band / pipeline machinery, by hand on a GPU box (not collected by pytest):
    python tests/fuzz_sequences.py [--cases N] [--seed S]

Each case draws a frame size (1..400 x 1..300), a sequence of 1..14 frames (own time, recursion limit and blend flag per
frame), a band split (strip height, parts, part, compact or full-frame addressing), a pipeline, a sub-frame count, a number
of frames in flight, a number of frames per launch chain, whether colour is the only plane, and a batch setting, renders the sequence with ONE rt_render_sequence call into host planes pre-filled with a pattern, and
compares colour, depth and the primary-hit planes bit for bit with the CPU oracle's frame-by-frame accumulation of the
same rows; rows the band part does not own must keep the pattern."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--big", action="store_true", help="frames of 2000..3840 x 1200..2160 pixels (the sizes at which sequences are traced four frames per launch and blended in the kernel)")
    a = ap.parse_args()
    from oracle import oracle
    from raytracer_amd import api, dist as rdist, scenes
    from raytracer_amd.camera import Camera
    from raytracer_amd.params import Uniforms
    rng = np.random.default_rng(a.seed)
    rng_flight = np.random.default_rng(a.seed + 10007)       # (a generator of its own: the cases of a seed stay what they were)
    oc = scenes.default_octree()
    sky = rng.integers(0, 256, size=(6, 32, 32, 4), dtype=np.uint8)
    osc = oracle.OracleScene(oc.as_dict(), sky)
    sc = api.Scene.from_octree(oc); sc.set_skybox(sky)
    bad = 0
    for case in range(a.cases):
        w, h = int(rng.integers(1, 401)), int(rng.integers(1, 301))
        if rng.uniform() < 0.15:
            w, h = int(rng.integers(600, 1400)), int(rng.integers(400, 800))     # large enough for several sub-frames
        if a.big:
            w, h = int(rng.integers(2000, 3841)), int(rng.integers(1200, 2161))
        n = int(rng.integers(1, 10 if a.big else 15))
        band_rows = int(rng.choice([1, 2, 7, 16, 16, 16, 32]))
        parts = int(rng.choice([1, 1, 2, 3, 5]))
        part = int(rng.integers(0, parts))
        compact = bool(rng.integers(0, 2)) and parts > 1
        pipe = str(rng.choice(["auto", "megakernel", "wavefront", "persistent"]))
        sub, batch = int(rng.integers(1, 5)), int(rng.choice([0, 0, 1, 2, 3, 4, 8]))
        a.eye = (float(rng.uniform(-4, 4)), float(rng.uniform(0.3, 4)), float(rng.uniform(-4, 4)))
        cam = Camera(eye=a.eye, at=(0.0, 1.0, 0.0), aspect=w / h)
        base_limit = int(rng.integers(1, 6))
        start = int(rng.choice([1, 1, int(rng.integers(2, 1000))]))
        us, blends = [], []
        for k in range(n):
            lim = base_limit if rng.uniform() < 0.85 else int(rng.integers(0, 7))
            us.append(Uniforms.from_camera(cam, time=float(rng.uniform(0, 50)), max_recursion_limit=lim).pack())
            blends.append(bool(rng.uniform() < 0.85))
        flight = int(rng_flight.choice([0, 0, 1, 2, 3, 4, 8]))
        stack = int(rng_flight.choice([0, 0, 1, 2, 3, 8]))
        color_only = bool(rng_flight.uniform() < 0.5)       # (frames only share launches when colour is the only plane asked for)
        sc.set_pipeline(pipe); sc.set_subframes(sub); sc.set_frames_in_flight(flight); sc.set_frames_per_launch(stack); sc.set_sequence_batch(batch)
        frames = [api.make_frame(us[k], w, h, accum_index=start + k, blend=blends[k], band=(band_rows, parts, part), compact=compact) for k in range(n)]
        rows = api.band_row_count(frames[0]) if compact else h
        ys = rdist.local_to_frame_rows(h, band_rows, parts, part)                 # frame row of every local row (-1: beyond the frame)
        own = ys[ys >= 0]
        pattern = dict(color=0.125, depth=0.75, kind=77, slot=123456, t=-3.0, nrays=55)
        got = dict(color=np.full((rows, w, 4), pattern["color"], np.float32), depth=np.full((rows, w), pattern["depth"], np.float32), kind=np.full((rows, w), pattern["kind"], np.uint32),
                   slot=np.full((rows, w), pattern["slot"], np.uint32), t=np.full((rows, w), pattern["t"], np.float32), nrays=np.full((rows, w), pattern["nrays"], np.uint32))
        tg = api.rt_target()
        tg.memory, tg.sync = api.RT_MEM_HOST, 1
        tg.color = got["color"].ctypes.data
        if color_only:
            got = {"color": got["color"]}
        else:
            tg.depth, tg.hit_kind, tg.hit_slot, tg.hit_t, tg.n_rays = (got[k].ctypes.data for k in ("depth", "kind", "slot", "t", "nrays"))
        sc.render_sequence_raw(frames, tg)
        # the oracle, frame by frame over the full frame
        acc = np.full((h, w, 4), pattern["color"], np.float32)
        last = None
        for k in range(n):
            last = osc.render(us[k], w, h, accum_index=start + k, blend=blends[k], rgba=acc)
            acc = last["rgba"]
        want = dict(color=acc, depth=last["depth"], kind=last["kind"], slot=last["slot"], t=last["t"], nrays=last["nrays"])
        if int(us[-1][19:20].view(np.uint32)[0]) == 0:
            want["t"] = np.full((h, w), np.inf, np.float32)                        # no segment: the distance plane holds +inf
        diffs = {}
        for key in got:
            if compact:
                g = got[key][:len(ys)][ys >= 0]; wv = want[key][own]
                rest_ok = True
            else:
                g = got[key][own]; wv = want[key][own]
                others = np.ones(h, bool); others[own] = False
                rest_ok = bool((got[key][others] == np.array(pattern[key], got[key].dtype)).all())
            d = int((np.ascontiguousarray(g).view(np.uint32) != np.ascontiguousarray(wv).astype(g.dtype).view(np.uint32)).sum())
            if d or not rest_ok:
                diffs[key] = (d, "foreign rows touched" if not rest_ok else "")
        if diffs:
            bad += 1
            print(f"MISMATCH case {case} seed {a.seed}: {w}x{h} n={n} band=({band_rows},{parts},{part}) compact={compact} pipe={pipe} sub={sub} flight={flight} stack={stack} color_only={color_only} batch={batch} start={start} "
                  f"limits={[int(u[19:20].view(np.uint32)[0]) for u in us]} blends={[int(b) for b in blends]}: {diffs}", flush=True)
        if case % 20 == 19:
            print(f"{case + 1} cases, {bad} mismatching", flush=True)
    sc.close()
    print(f"done: {a.cases} cases, {bad} mismatching")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
