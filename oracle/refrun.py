"""TEST INFRASTRUCTURE: run the reference's own fragment shader on Mesa llvmpipe.

Drives oracle/_ref/ref_runner (see ref_runner.c).  Works only where /root/reference exists;
it is used to generate tests/golden/ and to validate oracle/rt_oracle.c, never at GPU-test time.

"Tap" renders export per-pixel internals of the reference shader (primary hit kind / slot / t,
number of path segments, RNG seed).  The tapped shader text is derived from the reference file
IN MEMORY by a handful of anchored insertions and written only to a temporary file outside the
repository; the anchors are the minimal fragments of the reference's lines needed to locate them
(/root/reference/app/assets/ray_tracer.frag:203,527-560,573,670).
"""
from __future__ import annotations

import os
import re
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
RUNNER = os.path.join(HERE, "_ref", "ref_runner")
REF_FRAG = "/root/reference/app/assets/ray_tracer.frag"

# kinds exported by the "hit" tap
KIND_MISS, KIND_SPHERE, KIND_TRIANGLE, KIND_CYLINDER, KIND_PORTAL_1, KIND_PORTAL_2 = 0, 1, 2, 3, 4, 5


def available() -> bool:
    return os.path.exists(RUNNER) and os.path.exists(REF_FRAG)


def build() -> None:
    subprocess.run(["make", "-C", HERE, "ref"], check=True, capture_output=True)


def _tapped_frag(variant: str) -> str:
    src = open(REF_FRAG).read()
    lines = src.split("\n")
    out = []
    n_ret = 0
    # per-return tap payloads, in the order the returns appear in FindIntersection (frag:527-560)
    ret_taps = [
        (KIND_PORTAL_1, "0u", "portal_1_t"),
        (KIND_PORTAL_2, "0u", "portal_2_t"),
        (KIND_MISS, "0u", "closest_distance"),
        (KIND_CYLINDER, "0u", "closest_distance"),
        (KIND_TRIANGLE, "closest_triangle_start", "closest_distance"),
        (KIND_SPHERE, "closest_i", "closest_distance"),
    ]
    done = {"glob": False, "count": False, "out": False}
    for ln in lines:
        if not done["glob"] and re.match(r"\s*float min3\(", ln):
            out.append("float tap_kind = 0.0; float tap_lo = 0.0; float tap_hi = 0.0; float tap_t = 0.0; float tap_n = 0.0; float tap_seed = 0.0;")
            done["glob"] = True
        if re.search(r"return HitInfo\(", ln):
            k, idx, t = ret_taps[n_ret]
            n_ret += 1
            indent = re.match(r"\s*", ln).group(0)
            out.append(f"{indent}if (tap_n == 0.0) {{ tap_kind = {float(k)}; tap_lo = float(({idx}) & 0xFFFFu); tap_hi = float(({idx}) >> 16u); tap_t = {t}; }}")
        if re.search(r"fs_out_col\s*=\s*vec4\(color", ln):
            if variant == "hit":
                out.append("    fs_out_col = vec4(tap_kind * 16.0 + tap_n, tap_lo, tap_hi, tap_t);")
            elif variant == "seed":
                out.append("    fs_out_col = vec4(tap_kind * 16.0 + tap_n, seed, depth, tap_t);")
            else:
                raise ValueError(variant)
            done["out"] = True
            continue
        out.append(ln)
        if re.search(r"=\s*FindIntersection\(ray\);", ln):
            out.append("        tap_n += 1.0;")
            done["count"] = True
    if n_ret != len(ret_taps) or not all(done.values()):
        raise RuntimeError(f"tap anchors not found in the reference shader: returns={n_ret} {done}")
    return "\n".join(out)


def run_reference(scene: dict, uniforms70: np.ndarray, width: int, height: int, *, sky=None, times=None,
                  tap: str | None = None, blend: bool = True, bench: int = 0, threads: int | None = None) -> dict:
    """Render with the reference shader.  scene: dict(vertices[n,4] f32, normals[n,4] f32, tri_refs[m,4] u32,
    nodes[k] u32, bounds_min[3], bounds_max[3]).  sky: uint8 array [6, h, w, 4] or None.
    Returns dict(rgba=[H,W,4] f32 (row 0 = bottom), depth=[H,W] f32, plus tap fields)."""
    if not available():
        raise RuntimeError("reference runner not available (needs /root/reference and `make -C oracle ref`)")
    with tempfile.TemporaryDirectory(prefix="rtref_") as td:
        def w(name, arr):
            p = os.path.join(td, name)
            np.ascontiguousarray(arr).tofile(p)
            return p
        args = [RUNNER, "--quiet", "--w", str(width), "--h", str(height), "--out", os.path.join(td, "o")]
        args += ["--ssbo0", w("v", np.asarray(scene["vertices"], np.float32)), "--ssbo1", w("n", np.asarray(scene["normals"], np.float32)),
                 "--ssbo2", w("t", np.asarray(scene["tri_refs"], np.uint32)), "--ssbo3", w("k", np.asarray(scene["nodes"], np.uint32))]
        args += ["--bounds", w("b", np.concatenate([np.asarray(scene["bounds_min"], np.float32), np.asarray(scene["bounds_max"], np.float32)]))]
        args += ["--uniforms", w("u", np.asarray(uniforms70, np.float32))]
        if sky is not None:
            sky = np.ascontiguousarray(sky, dtype=np.uint8)
            assert sky.ndim == 4 and sky.shape[0] == 6 and sky.shape[3] == 4
            args += ["--sky", w("s", sky), str(sky.shape[2]), str(sky.shape[1])]
        if times is not None:
            args += ["--times", w("tm", np.asarray(times, np.float32))]
        if tap is not None:
            p = os.path.join(td, "tap.frag")
            with open(p, "w") as f:
                f.write(_tapped_frag(tap))
            args += ["--frag", p, "--noblend"]
        elif not blend:
            args += ["--noblend"]
        if bench:
            args += ["--bench", str(bench)]
        env = dict(os.environ)
        if threads is not None:
            env["LP_NUM_THREADS"] = str(threads)
        r = subprocess.run(args, check=True, capture_output=True, text=True, env=env)
        rgba = np.fromfile(os.path.join(td, "o.rgba.f32"), dtype=np.float32).reshape(height, width, 4)
        depth = np.fromfile(os.path.join(td, "o.depth.f32"), dtype=np.float32).reshape(height, width)
    out = {"rgba": rgba, "depth": depth, "stdout": r.stdout}
    if tap is not None:
        kn = rgba[..., 0]
        out["kind"] = np.floor(kn / 16.0).astype(np.uint32)
        out["nrays"] = (kn - 16.0 * np.floor(kn / 16.0)).astype(np.uint32)
        out["t"] = rgba[..., 3].copy()
        if tap == "hit":
            out["slot"] = rgba[..., 1].astype(np.uint32) + (rgba[..., 2].astype(np.uint32) << 16)
        else:
            out["seed"] = rgba[..., 1].copy()
            out["shader_depth"] = rgba[..., 2].copy()
    return out


def run_probe(frag_text: str, ssbo0: np.ndarray, width: int, height: int, *, sky=None) -> np.ndarray:
    """Run an arbitrary probe fragment shader (own text, reads `ssbo0`, writes 4 floats) on llvmpipe."""
    with tempfile.TemporaryDirectory(prefix="rtprobe_") as td:
        pf = os.path.join(td, "p.frag")
        open(pf, "w").write(frag_text)
        pv = os.path.join(td, "p.vert")
        open(pv, "w").write("#version 430\nout vec2 vs_out_ndc_coord;\nvoid main(){ vec2 p = vec2(gl_VertexID == 1 ? 3.0 : -1.0, gl_VertexID == 2 ? 3.0 : -1.0); gl_Position = vec4(p, 0.0, 1.0); vs_out_ndc_coord = p; }\n")
        pi = os.path.join(td, "in")
        np.ascontiguousarray(ssbo0).tofile(pi)
        args = [RUNNER, "--quiet", "--w", str(width), "--h", str(height), "--vert", pv, "--frag", pf, "--ssbo0", pi, "--noblend", "--out", os.path.join(td, "o")]
        if sky is not None:
            ps = os.path.join(td, "s")
            np.ascontiguousarray(sky, dtype=np.uint8).tofile(ps)
            args += ["--sky", ps, str(sky.shape[2]), str(sky.shape[1])]
        subprocess.run(args, check=True, capture_output=True, text=True)
        return np.fromfile(os.path.join(td, "o.rgba.f32"), dtype=np.float32).reshape(height, width, 4)


def overlay_uniform_words(view_proj, worlds, world_its, light_dir, la, ld) -> np.ndarray:
    """The overlay pass's uniforms as the runner reads them: viewProj[16] (column-major), light_dir[3], La[3], Ld[3], 1 pad word,
    then world[16] + worldIT[16] per draw (App.cpp:152-163)."""
    head = np.concatenate([np.asarray(view_proj, np.float32).reshape(16), np.asarray(light_dir, np.float32), np.asarray(la, np.float32), np.asarray(ld, np.float32), np.zeros(1, np.float32)])
    inst = [np.concatenate([np.asarray(w, np.float32).reshape(16), np.asarray(wi, np.float32).reshape(16)]) for w, wi in zip(worlds, world_its)]
    return np.concatenate([head] + inst).astype(np.float32)


def run_overlay(vertices: np.ndarray, indices: np.ndarray, overlay_words: np.ndarray, texture: np.ndarray, width: int, height: int, *,
                scene: dict | None = None, uniforms70=None, sky=None, times=None, frag_text: str | None = None, vert_text: str | None = None, present: bool = False) -> dict:
    """The reference's raster pass (its own two shader files unless probe texts are given) on llvmpipe: `vertices` [n,8] f32
    (position, normal, texcoord), `indices` u32, `texture` [h,w,4] u8 (row 0 = bottom, as uploaded).  With `scene` the ray-tracing
    draw runs first, as in the reference's frame; without it the pass draws over colour 0 / depth 1.  Returns rgba [H,W,4], depth [H,W]."""
    if not available():
        raise RuntimeError("reference runner not available (needs /root/reference and `make -C oracle ref`)")
    with tempfile.TemporaryDirectory(prefix="rtovl_") as td:
        def w(name, arr):
            p = os.path.join(td, name)
            np.ascontiguousarray(arr).tofile(p)
            return p
        args = [RUNNER, "--quiet", "--w", str(width), "--h", str(height), "--out", os.path.join(td, "o")]
        args += ["--overlay", w("ov", np.asarray(vertices, np.float32)), w("oi", np.asarray(indices, np.uint32)), w("ou", np.asarray(overlay_words, np.float32))]
        texture = np.ascontiguousarray(texture, np.uint8)
        args += ["--overlay-tex", w("ot", texture), str(texture.shape[1]), str(texture.shape[0])]
        for opt, text in (("--overlay-frag", frag_text), ("--overlay-vert", vert_text)):
            if text is not None:
                p = os.path.join(td, opt.strip("-") + ".glsl")
                open(p, "w").write(text)
                args += [opt, p]
        if scene is None:
            args += ["--no-rt", "--noblend"]
        else:
            args += ["--ssbo0", w("v", np.asarray(scene["vertices"], np.float32)), "--ssbo1", w("n", np.asarray(scene["normals"], np.float32)),
                     "--ssbo2", w("t", np.asarray(scene["tri_refs"], np.uint32)), "--ssbo3", w("k", np.asarray(scene["nodes"], np.uint32))]
            args += ["--bounds", w("b", np.concatenate([np.asarray(scene["bounds_min"], np.float32), np.asarray(scene["bounds_max"], np.float32)]))]
            args += ["--uniforms", w("u", np.asarray(uniforms70, np.float32))]
            if sky is not None:
                sky = np.ascontiguousarray(sky, dtype=np.uint8)
                args += ["--sky", w("s", sky), str(sky.shape[2]), str(sky.shape[1])]
            if times is not None:      # one whole frame (ray-tracing draw blended 1/(k+1), then the raster pass) per entry
                args += ["--times", w("tm", np.asarray(times, np.float32))]
        if present:
            args += ["--present"]
        subprocess.run(args, check=True, capture_output=True, text=True)
        rgba = np.fromfile(os.path.join(td, "o.rgba.f32"), dtype=np.float32).reshape(height, width, 4)
        depth = np.fromfile(os.path.join(td, "o.depth.f32"), dtype=np.float32).reshape(height, width)
        out = {"rgba": rgba, "depth": depth}
        if present:
            out["present"] = np.fromfile(os.path.join(td, "o.present.u8"), dtype=np.uint8).reshape(height, width, 4)
    return out
