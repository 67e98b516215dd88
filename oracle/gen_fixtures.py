"""TEST INFRASTRUCTURE: generate the data fixtures under tests/golden/ in the build container.

Everything written is DATA: scene inputs and the outputs of the reference's own shader executed on Mesa llvmpipe
(oracle/_ref/ref_runner reads the shader text from /root/reference at run time; nothing of it is stored).

    python -m oracle.gen_fixtures            # regenerate all fixtures (needs /root/reference and llvmpipe)

Fixtures
  bunny_mesh.npz          positions/faces of the reference's assets/stanford_bunny.obj (scene input)
  ref_<view>.npz          per view: uniforms, sky, size; reference colour (rgb f32), gl_FragDepth, and taps of the
                          primary hit (kind, triangle-reference slot / sphere index, t), segment count and final RNG seed
  ref_math.npz            llvmpipe's sin/cos/pow and cubemap lookups on probe inputs (own probe shader text)
  ref_sky_sizes.npz       cubemap lookups at face sizes 9, 30 and 100 (own probe shader text)
  ref_accum.npz           4 progressively blended frames (time varies), final colour; ref_accum8.npz: 8 frames
  refscene_<name>.npz     the same record for scenes other than the bunny, stored WITH their scene arrays (vertices, normals,
                          triangle references, node words, bounds): a cluster of small meshes carrying all seven materials
                          on triangles, and the reference's quad mesh assets/suzanne.obj (as triangulated by oracle.parse_obj)
  ref_ndc_640x512.npz     the interpolated NDC varying at the reference's own window size (not a power of two, so it
                          differs from the closed form): int32 ulp deltas, plus reference taps/colour for a band of rows
  ref_ndc_1280x720.npz, ref_ndc_3840x2160.npz, ref_ndc_7680x4320.npz   the same at the benchmark frame sizes (int16 deltas, bands of rows)
  refsynth_<scene>_<view>.npz   reference records of the seeded synthetic scenes of raytracer_amd/scenes.py (100 000 primitives
                          in a depth-8 octree; 1.28 M triangles in a depth-18 octree), stored WITHOUT the scene arrays: the
                          tests rebuild the scene from its seed and check the SHA-256 digests kept here
  refscene_deep.npz       a tree 26 levels deep (scenes.deep_test_scene), with its arrays
  refscene_planes_*.npz   scenes.plane_aligned_scene (sheet vertices, edges and corners exactly on the octree's cell planes) from eyes on
                          cell planes, at odd frame sizes with the rasteriser's NDC stored as ulp deltas
  ref_fullframe_digests.json   SHA-256 digests of the reference's WHOLE frames at 1280x720 and 3840x2160 (bunny and both synthetic scenes)
                          and llvmpipe's frame times for them
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import hashlib  # noqa: E402

from oracle import oracle, refrun  # noqa: E402
from raytracer_amd.camera import Camera, Portals, normalize  # noqa: E402
from raytracer_amd.params import Uniforms  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
BUNNY_OBJ = "/root/reference/app/assets/stanford_bunny.obj"

VIEWS = {
    "default": dict(cam=Camera(aspect=1.0), size=(256, 256)),
    "wide": dict(cam=Camera(aspect=2.0), size=(256, 128)),
    "portal": dict(cam=Camera(eye=(3.0, 2.0, 3.0), at=(7.0, 2.0, 3.0), aspect=1.0), size=(128, 128)),
    "portal2": dict(cam=Camera(eye=(9.5, 2.2, 6.5), at=(7.0, 1.5, 4.0), aspect=1.0), size=(128, 128)),
    "glass": dict(cam=Camera(eye=(-2.0, 2.5, 3.0), at=(-4.0, 2.0, 0.0), aspect=1.0), size=(128, 128)),
    "close": dict(cam=Camera(eye=(0.5, 2.5, 1.0), at=(-0.3, 1.9, 0.0), aspect=1.0), size=(128, 128)),
    "time": dict(cam=Camera(aspect=1.0), size=(128, 128), kw=dict(time=3.7)),
    "topdown": dict(cam=Camera(eye=(0.1, 9.0, 0.1), at=(0.0, 0.0, 0.0), aspect=1.0), size=(128, 128)),
    "maxrec2": dict(cam=Camera(aspect=1.0), size=(128, 128), kw=dict(max_recursion_limit=2)),
    "maxrec8": dict(cam=Camera(aspect=1.0), size=(128, 128), kw=dict(max_recursion_limit=8, blur_amount=0.01)),
    # edge cases: no segment at all, primary only, a camera 96 units out (beyond the distance up to which the product culls
    # spheres), a camera inside a big sphere, a large time value, a strong lens blur
    "maxrec0": dict(cam=Camera(aspect=1.0), size=(64, 64), kw=dict(max_recursion_limit=0)),
    "maxrec1": dict(cam=Camera(aspect=1.0), size=(128, 128), kw=dict(max_recursion_limit=1)),
    "far": dict(cam=Camera(eye=(60.0, 45.0, 60.0), at=(0.0, 1.0, 0.0), fovy=0.2, aspect=1.0), size=(128, 128)),
    "inside_sphere": dict(cam=Camera(eye=(-4.0, 2.0, 0.3), at=(0.0, 2.0, 0.0), aspect=1.0), size=(128, 128)),
    "bigtime": dict(cam=Camera(aspect=1.0), size=(128, 128), kw=dict(time=12345.678)),
    "blur": dict(cam=Camera(aspect=1.0), size=(128, 128), kw=dict(blur_amount=0.2)),
    # other clip planes: gl_FragDepth is a function of z_near / z_far (frag:352-354), hits beyond z_far clamp to 1
    "depthrange": dict(cam=Camera(aspect=1.0, z_near=0.5, z_far=6.0), size=(128, 128)),
    # near planes far out: gl_FragDepth values over the whole of [0, 1] (with the default planes they stay above 0.78), which
    # is where the 24-bit depth attachment's rounding of small values shows
    "znear03": dict(cam=Camera(aspect=1.0, z_near=0.3, z_far=1000.0), size=(128, 128)),
    "znear1": dict(cam=Camera(eye=(3.0, 1.5, 0.5), at=(-0.3, 1.9, 0.0), aspect=1.0, z_near=1.0, z_far=40.0), size=(128, 128)),
    "znear2": dict(cam=Camera(eye=(-2.0, 2.5, 3.0), at=(0.0, 1.0, 0.0), aspect=1.0, z_near=2.5, z_far=300.0), size=(128, 128)),
    # portals away from their default places, neither axis-aligned: both difference matrices are general rotations
    "portal_custom": dict(cam=Camera(eye=(2.0, 2.0, 3.5), at=(0.0, 1.5, 0.0), aspect=1.0), size=(128, 128),
                          kw=dict(portals=Portals(position_1=(0.9, 1.9, 2.2), direction_1=tuple(normalize((0.3, 0.0, 1.0)).astype(np.float32).tolist()),
                                                  position_2=(-2.5, 1.6, -1.0), direction_2=tuple(normalize((1.0, 0.2, 0.1)).astype(np.float32).tolist()), width=1.4, height=1.8))),
}


def fixture_sky(seed: int = 11, size: int = 16) -> np.ndarray:
    return np.random.default_rng(seed).integers(0, 256, size=(6, size, size, 4), dtype=np.uint8)


def bunny_scene() -> dict:
    z = np.load(os.path.join(GOLDEN, "bunny_mesh.npz"))
    T = np.diag([20.0, 20.0, 20.0, 1.0]).astype(np.float32)
    v, n, t = oracle.mesh_build(z["positions"], None, z["faces"], 1, T)
    o = oracle.octree_build(v, t, 18, 10, 6, 6)
    return dict(vertices=v, normals=n, tri_refs=o["tri_refs"], nodes=o["nodes"], bounds_min=o["bounds_min"], bounds_max=o["bounds_max"])


def reference_record(scene, u, w, h, sky, rows=None) -> dict:
    col = refrun.run_reference(scene, u, w, h, sky=sky)
    hit = refrun.run_reference(scene, u, w, h, sky=sky, tap="hit")
    sd = refrun.run_reference(scene, u, w, h, sky=sky, tap="seed")
    assert np.array_equal(hit["kind"], sd["kind"]) and np.array_equal(hit["nrays"], sd["nrays"])
    rec = dict(uniforms=np.asarray(u, np.float32), sky=sky, size=np.array([w, h], np.int32),
               rgb=col["rgba"][..., :3].copy(), alpha_max=np.float32(np.abs(col["rgba"][..., 3]).max()), depth=sd["shader_depth"],
               attachment_depth=col["depth"], kind=hit["kind"].astype(np.uint8), slot=hit["slot"], t=hit["t"], nrays=hit["nrays"].astype(np.uint8), seed=sd["seed"])
    if rows is not None:
        a, b = rows
        for k in ("rgb", "depth", "attachment_depth", "kind", "slot", "t", "nrays", "seed"):
            rec[k] = rec[k][a:b].copy()
        rec["rows"] = np.array([a, b], np.int32)
    return rec


INTERP_FRAG = """#version 430
in vec2 vs_out_ndc_coord;
out vec4 fs_out_col;
void main() { fs_out_col = vec4(vs_out_ndc_coord, gl_FragCoord.xy); }
"""


def capture_ndc(w: int, h: int):
    """The rasteriser-interpolated varying of the reference's fullscreen triangle (its own vertex shader)."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        pf = os.path.join(td, "i.frag")
        open(pf, "w").write(INTERP_FRAG)
        subprocess.run([refrun.RUNNER, "--quiet", "--w", str(w), "--h", str(h), "--frag", pf, "--noblend", "--out", os.path.join(td, "o")], check=True, capture_output=True)
        r = np.fromfile(os.path.join(td, "o.rgba.f32"), np.float32).reshape(h, w, 4)
    return r[..., 0].copy(), r[..., 1].copy()


def closed_form_ndc(w: int, h: int):
    x = (np.float32(2.0) * (np.arange(w, dtype=np.float32) + np.float32(0.5)) / np.float32(w) - np.float32(1.0)).astype(np.float32)
    y = (np.float32(2.0) * (np.arange(h, dtype=np.float32) + np.float32(0.5)) / np.float32(h) - np.float32(1.0)).astype(np.float32)
    return np.tile(x, (h, 1)), np.tile(y[:, None], (1, w))


def ulp_delta(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Signed distance in representable floats from b to a (monotone integer mapping of the bit patterns)."""
    def key(f):
        i = np.ascontiguousarray(f, np.float32).view(np.int32).astype(np.int64)
        return np.where(i < 0, -(i & 0x7FFFFFFF), i)
    return key(a) - key(b)


def apply_ulp_delta(b: np.ndarray, d: np.ndarray) -> np.ndarray:
    i = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    k = np.where(i < 0, -(i & 0x7FFFFFFF), i) + d.astype(np.int64)
    out = np.where(k < 0, (-k) | 0x80000000, k).astype(np.uint32)
    return out.view(np.float32)


MATH_FRAG = """#version 430
layout(std430, binding = 0) buffer B { readonly vec4 xin[]; };
uniform samplerCube skyboxTexture;
out vec4 fs_out_col;
void main() {
  uint i = uint(gl_FragCoord.y) * 64u + uint(gl_FragCoord.x);
  vec4 v = xin[i];
  if (gl_FragCoord.y < 32.0) fs_out_col = vec4(sin(v.x), cos(v.x), pow(v.y, 1./3.), pow(v.y, 5.0));
  else fs_out_col = texture(skyboxTexture, v.xyz);
}
"""


def math_fixture(sky):
    """Element-wise probes of the built-ins the shader uses (own probe text): rows 0-31 sin/cos/pow, rows 32-63 cubemap lookups."""
    rng = np.random.default_rng(5)
    x = np.zeros((64 * 64, 4), np.float32)
    n = 32 * 64
    x[:n, 0] = np.concatenate([np.linspace(0, 6.2831855, n - 64, dtype=np.float32), rng.uniform(-6.3, 6.3, 64).astype(np.float32)])
    x[:n, 1] = np.concatenate([[0.0, 1.0, 0.5, 2.0], rng.random(n - 4, dtype=np.float32) * np.float32(2.0)]).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    special = np.array([[np.nan] * 3, [1, 1, 1], [1, 1, 0.5], [0.5, 1, 1], [-1, 1, 1], [1, -1, -1], [0.25, 0.5, -1], [-0.5, -0.25, 1], [np.inf, 0, 0], [1, 0, 0], [0, -1, 0], [0, 0, -1]], np.float32)
    d[:len(special)] = special
    d[12] = np.array([0xFFC00000] * 3, np.uint32).view(np.float32)   # the quiet NaN an x86 invalid operation produces
    x[n:, :3] = d
    out = refrun.run_probe(MATH_FRAG, x, 64, 64, sky=sky).reshape(-1, 4)
    return dict(x=x[:n, 0].copy(), p=x[:n, 1].copy(), sin=out[:n, 0].copy(), cos=out[:n, 1].copy(), cbrt=out[:n, 2].copy(), pow5=out[:n, 3].copy(),
                dirs=d, sky_rgb=out[n:, :3].copy(), sky=sky)


def materials_scene() -> dict:
    """60 small icospheres and boxes around (0, 1.5, 0), materials 0..6 cycling: every material on mesh triangles."""
    from raytracer_amd import scenes
    meshes = scenes.synthetic_primitives(60, seed=3, sphere_subdiv=1, extent=1.5, centre=(0.0, 1.5, 0.0), size_range=(0.1, 0.35), materials=(0, 1, 2, 3, 4, 5, 6))
    verts, norms, tris, base = [], [], [], 0
    for m in meshes:                                  # merged as Octree.cpp:221-233 merges its meshes
        verts.append(m.m_vertecies); norms.append(m.m_normals)
        t = m.m_triangles.copy(); t[:, :3] += base
        tris.append(t); base += len(m.m_vertecies)
    v, n, t = np.concatenate(verts), np.concatenate(norms), np.concatenate(tris)
    o = oracle.octree_build(v, t, 10, 10, 6, 6)
    return dict(vertices=v, normals=n, tri_refs=o["tri_refs"], nodes=o["nodes"], bounds_min=o["bounds_min"], bounds_max=o["bounds_max"])


def suzanne_scene(material_id: int) -> dict:
    pos, nrm, faces = oracle.parse_obj("/root/reference/app/assets/suzanne.obj")
    T = np.eye(4, dtype=np.float32)          # the mesh sits around (-2.5, 1.25, 4.1) in its own coordinates
    v, n, t = oracle.mesh_build(pos, nrm, faces, material_id, T)
    o = oracle.octree_build(v, t, 18, 10, 6, 6)
    return dict(vertices=v, normals=n, tri_refs=o["tri_refs"], nodes=o["nodes"], bounds_min=o["bounds_min"], bounds_max=o["bounds_max"])


# (power-of-two frame sizes: there the rasteriser's interpolated NDC equals the closed form, see ref_ndc_640x512.npz for the other case)
SCENE_VIEWS = {
    "materials_a": dict(scene="materials", cam=Camera(eye=(2.5, 2.5, 2.5), at=(0.0, 1.5, 0.0), aspect=1.0), size=(128, 128)),
    "materials_b": dict(scene="materials", cam=Camera(eye=(1.3, 2.3, 1.5), at=(0.0, 1.5, 0.0), fovy=1.2, aspect=2.0), size=(256, 128), kw=dict(max_recursion_limit=8)),
    "suzanne_glass": dict(scene="suzanne5", cam=Camera(eye=(-1.3, 1.9, 6.4), at=(-2.5, 1.25, 4.1), aspect=1.0), size=(128, 128), kw=dict(max_recursion_limit=8)),
    "empty": dict(scene="empty", cam=Camera(aspect=1.0), size=(128, 128)),
    "suzanne_metal": dict(scene="suzanne2", cam=Camera(eye=(-4.6, 2.4, 5.9), at=(-2.5, 1.25, 4.1), aspect=1.0), size=(128, 128)),
    # a tree 26 levels deep (the reference's own limit is 18): the traversal stacks of the deep-tree kernels
    "deep": dict(scene="deep", cam=Camera(eye=(0.9, 4.2, 0.8), at=(0.0, 3.6, 0.0), fovy=0.8, aspect=1.0), size=(128, 128), kw=dict(max_recursion_limit=6)),
    # scenes.plane_aligned_scene: sheet vertices, edges and corners exactly on the octree's cell planes.  An oblique view; then views
    # whose eye sits ON cell planes and looks along an axis with no lens blur, at ODD frame sizes (the rasteriser's interpolated NDC
    # is captured and stored as for ref_ndc_*.npz), so that the centre row and column of rays run inside cell planes, through sheet
    # edges and vertices; one eye is a lattice point inside the root box.
    "planes_oblique": dict(scene="planes", cam=Camera(eye=(6.5, 7.0, 5.5), at=(0.0, 3.5, 0.0), aspect=1.0), size=(256, 256), kw=dict(max_recursion_limit=5)),
    "planes_axis_z": dict(scene="planes", cam=Camera(eye=(0.0, 3.5, -7.0), at=(0.0, 3.5, 0.0), fovy=1.0, aspect=1.0), size=(257, 257), kw=dict(max_recursion_limit=4, blur_amount=0.0)),
    "planes_axis_x": dict(scene="planes", cam=Camera(eye=(6.5, 4.0, 0.5), at=(0.0, 4.0, 0.5), fovy=0.9, aspect=255.0 / 129.0), size=(255, 129), kw=dict(max_recursion_limit=4, blur_amount=0.0)),
    "planes_down": dict(scene="planes", cam=Camera(eye=(0.5, 7.75, -0.5), at=(0.5, 0.0, -0.5), up=(0.0, 0.0, 1.0), fovy=1.1, aspect=1.0), size=(193, 193), kw=dict(max_recursion_limit=3, blur_amount=0.0)),
    "planes_inside": dict(scene="planes", cam=Camera(eye=(0.0, 5.0, 0.0), at=(1.0, 4.0, 1.0), fovy=1.3, aspect=1.0), size=(128, 128), kw=dict(max_recursion_limit=5, blur_amount=0.0)),
    # a view the randomised runs found (tests/fuzz_parity.py --scene planes, seed 61, view 5820): eye and target are lattice points, and
    # the primary ray of pixel (19, 84) lies IN the plane of a sheet triangle seven units away -- its Moeller-Trumbore determinant is
    # rounding noise (2e-9), its numerator exactly zero, and the shader accepts it at distance 0 when it tests it before anything nearer
    "planes_coplanar": dict(scene="planes", cam=Camera(eye=(6.5, 1.0, -5.5), at=(2.75, 2.75, -2.5), fovy=0.5, aspect=2.0), size=(256, 128),
                            kw=dict(time=17.220732080302838, max_recursion_limit=6, blur_amount=0.0)),
}


def scene_fixtures():
    """refscene_<name>.npz: reference records of scenes other than the bunny, with the scene arrays inside."""
    sky = fixture_sky()
    # no mesh at all: a root leaf without triangles in a box of zero size (never entered)
    empty = dict(vertices=np.zeros((1, 4), np.float32), normals=np.zeros((1, 4), np.float32), tri_refs=np.zeros((1, 4), np.uint32), nodes=np.zeros(2, np.uint32),
                 bounds_min=np.zeros(3, np.float32), bounds_max=np.zeros(3, np.float32))
    from raytracer_amd import scenes
    built = {"materials": materials_scene, "suzanne5": lambda: suzanne_scene(5), "suzanne2": lambda: suzanne_scene(2), "empty": lambda: empty,
             "deep": lambda: scenes.deep_test_scene().as_dict(), "planes": lambda: scenes.plane_aligned_scene().as_dict()}
    only = sys.argv[2:] if len(sys.argv) > 2 else None
    for name, v in SCENE_VIEWS.items():
        if only and name not in only:
            continue
        scene = built[v["scene"]]()
        w, h = v["size"]
        u = Uniforms.from_camera(v["cam"], **v.get("kw", {})).pack()
        rec = reference_record(scene, u, w, h, sky)
        rec.update({"scene_" + k: a for k, a in scene.items()})
        if (w & (w - 1)) or (h & (h - 1)):       # not powers of two: the rasteriser's NDC differs from the closed form (see ndc_fixture)
            nx, ny = capture_ndc(w, h)
            cx, cy = closed_form_ndc(w, h)
            dx, dy = ulp_delta(nx, cx), ulp_delta(ny, cy)
            # (the centre pixel of an odd size has the closed form 0 and an interpolated value of about 1e-8: a distance of
            # hundreds of millions of representable floats, hence 32 bits)
            assert np.abs(dx).max() < 2 ** 31 and np.abs(dy).max() < 2 ** 31
            assert np.array_equal(apply_ulp_delta(cx, dx).view(np.uint32), nx.view(np.uint32)) and np.array_equal(apply_ulp_delta(cy, dy).view(np.uint32), ny.view(np.uint32))
            rec["ndc_dx"], rec["ndc_dy"] = dx.astype(np.int32), dy.astype(np.int32)
            print(name, "centre column ndc x:", nx[h // 2, w // 2], "centre row ndc y:", ny[h // 2, w // 2])
        np.savez_compressed(os.path.join(GOLDEN, f"refscene_{name}.npz"), **rec)
        tri = rec["kind"] == 2
        print(name, w, h, "kinds", np.bincount(rec["kind"].ravel(), minlength=6), "materials of primary triangle hits", np.unique(np.asarray(scene["tri_refs"]).reshape(-1, 4)[rec["slot"][tri], 3]).tolist(),
              "max segments", int(rec["nrays"].max()), os.path.getsize(os.path.join(GOLDEN, f"refscene_{name}.npz")))


def scene_digests(scene: dict) -> dict:
    """SHA-256 of the bytes of each scene array (vertices, normals, tri_refs, nodes as the octree builders return them)."""
    out = {}
    for k, dt in (("vertices", np.float32), ("normals", np.float32), ("tri_refs", np.uint32), ("nodes", np.uint32), ("bounds_min", np.float32), ("bounds_max", np.float32)):
        out[k] = hashlib.sha256(np.ascontiguousarray(scene[k], dt).tobytes()).hexdigest()
    return out


# Views of the synthetic scenes (primitives fill the cube |x|,|z| <= 4 (3), -1 (0) <= y <= 7 (6) around (0, 3, 0)): the
# benchmark's camera, which stands inside the cube; one from outside; one from the centre with the shorter path.
SYNTH_VIEWS = {
    "bench": dict(cam=Camera(aspect=1.0), size=(256, 256), kw=dict(max_recursion_limit=5)),
    "outside": dict(cam=Camera(eye=(10.0, 7.0, 10.0), at=(0.0, 3.0, 0.0), aspect=2.0), size=(256, 128), kw=dict(max_recursion_limit=5, time=0.25)),
    "centre": dict(cam=Camera(eye=(0.0, 3.0, 0.0), at=(1.0, 2.6, 0.3), aspect=1.0), size=(128, 128), kw=dict(max_recursion_limit=2)),
}


def synthetic_fixtures():
    """refsynth_<scene>_<view>.npz: the reference shader on the two seeded synthetic scenes of raytracer_amd/scenes.py."""
    from raytracer_amd import scenes
    sky = fixture_sky()
    for name, build in (("100k", scenes.synthetic_100k), ("1m", scenes.synthetic_1m)):
        oc = build()
        scene = oc.as_dict()
        dig = scene_digests(scene)
        for vname, v in SYNTH_VIEWS.items():
            w, h = v["size"]
            u = Uniforms.from_camera(v["cam"], **v.get("kw", {})).pack()
            rec = reference_record(scene, u, w, h, sky)
            rec.update({"sha256_" + k: np.array(d) for k, d in dig.items()})
            rec["n_ref"], rec["n_words"], rec["max_depth"] = np.int64(len(scene["tri_refs"])), np.int64(len(scene["nodes"])), np.int64(oc.max_depth)
            path = os.path.join(GOLDEN, f"refsynth_{name}_{vname}.npz")
            np.savez_compressed(path, **rec)
            print("synthetic", name, vname, w, h, "kinds", np.bincount(rec["kind"].ravel(), minlength=6), "segments", np.bincount(rec["nrays"].ravel()), os.path.getsize(path))


def ndc_fixture(w: int, h: int, rows, **kw):
    """ref_ndc_<w>x<h>.npz: the rasteriser's interpolated NDC as ulp deltas against the closed form, and the reference's
    record of the bunny scene for a band of rows rendered with it."""
    scene, sky = bunny_scene(), fixture_sky()
    nx, ny = capture_ndc(w, h)
    cx, cy = closed_form_ndc(w, h)
    dx, dy = ulp_delta(nx, cx), ulp_delta(ny, cy)
    assert np.abs(dx).max() < 2 ** 15 and np.abs(dy).max() < 2 ** 15
    assert np.array_equal(apply_ulp_delta(cx, dx).view(np.uint32), nx.view(np.uint32)) and np.array_equal(apply_ulp_delta(cy, dy).view(np.uint32), ny.view(np.uint32))
    u = Uniforms.from_camera(Camera(aspect=w / h), **kw).pack()
    rec = reference_record(scene, u, w, h, sky, rows=rows)
    rec["ndc_dx"], rec["ndc_dy"] = dx.astype(np.int16), dy.astype(np.int16)
    path = os.path.join(GOLDEN, f"ref_ndc_{w}x{h}.npz")
    np.savez_compressed(path, **rec)
    print("ndc %dx%d: x differs on %.1f%% of pixels, y on %.1f%%; kinds in the band %s; %d bytes" % (w, h, 100 * (dx != 0).mean(), 100 * (dy != 0).mean(), np.bincount(rec["kind"].ravel(), minlength=6), os.path.getsize(path)))


def sky_size_fixture():
    """ref_sky_sizes.npz: cubemap lookups at face sizes that are not powers of two (the reference's faces are 900 x 900; the
    texel arithmetic scales coordinates by the size), on directions that include face edges and corners."""
    rng = np.random.default_rng(9)
    n = 32 * 64
    d = rng.standard_normal((n, 3)).astype(np.float32)
    d[:6] = np.array([[1, 1, 1], [1, -1, 1], [-1, 1, -1], [1, 0.999999, 0.5], [0.3, -1, -1], [-1, -1, -0.999999]], np.float32)
    x = np.zeros((64 * 64, 4), np.float32)
    x[32 * 64:, :3] = d
    rec = {"dirs": d}
    for size in (9, 30, 100):
        sky = np.random.default_rng(100 + size).integers(0, 256, size=(6, size, size, 4), dtype=np.uint8)
        out = refrun.run_probe(MATH_FRAG, x, 64, 64, sky=sky).reshape(-1, 4)
        rec[f"sky_{size}"] = sky
        rec[f"rgb_{size}"] = out[32 * 64:, :3].copy()
    # the reference's own face size; the faces are raytracer_amd.scenes.procedural_skybox(900) (seeded: not stored, only their digest)
    from raytracer_amd import scenes
    import hashlib
    sky = scenes.procedural_skybox(900)
    out = refrun.run_probe(MATH_FRAG, x, 64, 64, sky=sky).reshape(-1, 4)
    rec["sky_900_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(sky).tobytes()).digest(), np.uint8).copy()
    rec["rgb_900"] = out[32 * 64:, :3].copy()
    np.savez_compressed(os.path.join(GOLDEN, "ref_sky_sizes.npz"), **rec)
    print("sky sizes", os.path.getsize(os.path.join(GOLDEN, "ref_sky_sizes.npz")))


# Whole frames at the benchmark's sizes: too large to store (a 4K record is 200 MB), so their SHA-256 digests are kept instead.
FULLFRAME = {
    "bunny_720p_limit5": dict(scene="bunny", size=(1280, 720), kw=dict(max_recursion_limit=5)),
    "bunny_4k_limit2": dict(scene="bunny", size=(3840, 2160), kw=dict(max_recursion_limit=2)),
    "synthetic_100k_4k_limit5": dict(scene="synthetic_100k", size=(3840, 2160), kw=dict(max_recursion_limit=5)),
    "synthetic_1m_4k_limit5": dict(scene="synthetic_1m", size=(3840, 2160), kw=dict(max_recursion_limit=5)),
    "bunny_8k_limit5": dict(scene="bunny", size=(7680, 4320), kw=dict(max_recursion_limit=5)),      # the camera-path workload's frame size
}


def plane_digests(rec: dict) -> dict:
    """SHA-256 of every plane of a record as the tests hash them: rgb / depth / t float32, slot uint32, kind / nrays uint8."""
    out = {}
    for k, dt in (("rgb", np.float32), ("depth", np.float32), ("t", np.float32), ("slot", np.uint32), ("kind", np.uint8), ("nrays", np.uint8)):
        out[k] = hashlib.sha256(np.ascontiguousarray(rec[k], dt).tobytes()).hexdigest()
    return out


def fullframe_fixture(only=None):
    """ref_fullframe_digests.json: the reference shader's WHOLE frames at 1280x720, 3840x2160 and 7680x4320 -- the benchmark's camera on the
    bunny scene and on the two seeded synthetic scenes -- as digests of every plane, with the uniforms (bit patterns), the scene's
    digests and llvmpipe's frame time on this container's 8 threads (the CPU baseline of those workloads).  The rasteriser's
    interpolated NDC at these sizes is the table of ref_ndc_<w>x<h>.npz (captured again here and checked against it)."""
    import json
    from raytracer_amd import scenes
    sky = fixture_sky()
    path = os.path.join(GOLDEN, "ref_fullframe_digests.json")
    out = json.load(open(path)) if only and os.path.exists(path) else {}      # (named entries only: the others are kept)
    built = {}
    for name, v in FULLFRAME.items():
        if only and name not in only:
            continue
        if v["scene"] not in built:
            built[v["scene"]] = bunny_scene() if v["scene"] == "bunny" else {"synthetic_100k": scenes.synthetic_100k, "synthetic_1m": scenes.synthetic_1m}[v["scene"]]().as_dict()
        scene = built[v["scene"]]
        w, h = v["size"]
        nx, ny = capture_ndc(w, h)
        z = np.load(os.path.join(GOLDEN, f"ref_ndc_{w}x{h}.npz"))
        cx, cy = closed_form_ndc(w, h)
        assert np.array_equal(apply_ulp_delta(cx, z["ndc_dx"]).view(np.uint32), nx.view(np.uint32)) and np.array_equal(apply_ulp_delta(cy, z["ndc_dy"]).view(np.uint32), ny.view(np.uint32))
        u = Uniforms.from_camera(Camera(aspect=w / h), **v["kw"]).pack()
        rec = reference_record(scene, u, w, h, sky)
        timing = json.loads(refrun.run_reference(scene, u, w, h, sky=sky, bench=3, threads=8)["stdout"].strip().splitlines()[-1])
        out[name] = {"scene": v["scene"], "width": w, "height": h, "uniforms_bits": np.asarray(u, np.float32).view(np.uint32).tolist(),
                     "sky": "oracle.gen_fixtures.fixture_sky()", "scene_sha256": scene_digests(scene), "sha256": plane_digests(rec),
                     "kinds": np.bincount(rec["kind"].ravel(), minlength=6).tolist(), "rays": int(rec["nrays"].astype(np.int64).sum()),
                     "llvmpipe": {"threads": 8, "ms_best": timing["ms_best"], "ms_mean": timing["ms_mean"], "where": "the build container (8 vCPU)"}}
        print(name, out[name]["kinds"], out[name]["rays"], out[name]["llvmpipe"], flush=True)
    json.dump(out, open(path, "w"), indent=1)


def overlay_assets():
    """overlay_assets.npz: the mesh and texture of the reference's raster pass as its loaders hand them to GL (App.cpp:71-76):
    suzanne.obj through oracle.raster.load_obj, metal.png decoded to 8-bit RGBA and flipped (GLUtils.cpp:187-190)."""
    from oracle import raster
    from raytracer_amd.png import read_png_rgba
    v, i = raster.load_obj("/root/reference/app/assets/suzanne.obj")
    tex = read_png_rgba("/root/reference/app/assets/metal.png")[::-1].copy()
    np.savez_compressed(os.path.join(GOLDEN, "overlay_assets.npz"), vertices=v, indices=i, texture=tex)
    print("overlay assets", v.shape, i.shape, tex.shape, os.path.getsize(os.path.join(GOLDEN, "overlay_assets.npz")))
    return v, i, tex


def depth24_of(depth_readback: np.ndarray) -> np.ndarray:
    """The 24-bit values behind a GL_FLOAT read-back of the depth attachment (value / (2^24 - 1) as a float)."""
    return np.rint(depth_readback.astype(np.float64) * 16777215.0).astype(np.uint32)


def overlay_fixtures():
    """ref_overlay_soups.npz: triangle soups through the reference's raster shaders on llvmpipe (clipped by every frustum plane,
    orthographic and perspective, texture coordinates that magnify); ref_overlay_suzanne_*.npz: the reference's nine draws of its
    own mesh alone (over colour 0 / depth 1) and after the ray-tracing draw of the bunny scene (the whole frame of App::Render)."""
    from oracle import raster
    rng = np.random.default_rng(21)
    F = np.float32
    I = np.eye(4, dtype=F).reshape(-1)
    rec = {}
    tex = rng.integers(0, 256, (16, 8, 4), dtype=np.uint8)
    rec["texture"] = tex
    cam = Camera(eye=(0.3, 0.2, 1.0), at=(0, 0, 0), fovy=1.2, aspect=1.5, z_near=0.3, z_far=3.0)
    for k in range(8):
        n = 40
        v = np.zeros((3 * n, 8), F)
        spread, size = ((0.6, 0.39), (1.5, 1.2))[2 <= k < 6]
        c = rng.uniform(-spread, spread, (n, 1, 3)); v[:, :3] = (c + rng.uniform(-size, size, (n, 3, 3))).reshape(-1, 3)
        v[:, 3:6] = rng.uniform(-1, 1, (3 * n, 3))
        v[:, 6:8] = rng.uniform(0, 0.01, (3 * n, 2)) + rng.uniform(-2, 2, (n, 1, 2)).repeat(3, 1).reshape(-1, 2)
        if k == 5:      # vertices on pixel centres and corners: edges through sample points
            v[:, :2] = (rng.integers(0, 2 * 96 + 1, (3 * n, 2)) / F(96) - 1).astype(F) * np.array([1, 64 / 96], F)
        vp = cam.view_proj().ravel() if k % 2 else I
        words = raster.overlay_words(vp, [I], [I], (-1, -2, -0.5), (0.1, 0.2, 0.3), (0.9, 0.8, 0.7))
        if k >= 6:      # minified lookups: up to tens of texels per pixel, all mip levels, a texture whose sizes are not powers of two
            v[:, 6:8] = rng.uniform(-1, 1, (3 * n, 2)) * (2.0, 14.0)[k - 6] + rng.uniform(-2, 2, (n, 1, 2)).repeat(3, 1).reshape(-1, 2)
            rec[f"texture_{k}"] = rng.integers(0, 256, ((20, 12), (64, 32))[k - 6] + (4,), dtype=np.uint8)
        if k == 4:      # corners drawn from all over the soup: one texture coordinate for all
            v[:, 6:8] = (0.37, -1.21)
        idx = rng.permutation(3 * n).astype(np.uint32) if k == 4 else np.arange(3 * n, dtype=np.uint32)
        idx = idx[: 3 * (len(idx) // 3)]
        o = refrun.run_overlay(v, idx, words, rec.get(f"texture_{k}", tex), 96, 64)
        rec[f"vertices_{k}"] = v; rec[f"indices_{k}"] = idx; rec[f"words_{k}"] = words
        rec[f"rgba_{k}"] = o["rgba"]; rec[f"depth24_{k}"] = depth24_of(o["depth"])
        print("soup", k, "covered", int((rec[f"depth24_{k}"] != 0xFFFFFF).sum()))
    np.savez_compressed(os.path.join(GOLDEN, "ref_overlay_soups.npz"), **rec)
    print("overlay soups", os.path.getsize(os.path.join(GOLDEN, "ref_overlay_soups.npz")))

    v, i, mtex = overlay_assets()
    worlds, its = raster.reference_instances()
    # (the mesh file is not centred: its vertices span x -3.9..-1.1, y 0.3..2.2, z 3.3..5.0, so the nine draws are centred at (-2.5 + 4 i, 5.25 + 4 j, 4.1))
    # (power-of-two frame sizes: there the rasteriser's interpolated NDC of the ray-tracing draw equals the closed form the library uses by default)
    views = {"front": (Camera(eye=(-2.5, 5.25, 13.0), at=(-2.5, 5.25, 4.1), aspect=2.0), (256, 128)),
             "oblique": (Camera(eye=(6.0, 9.0, 10.0), at=(-2.5, 4.0, 4.1), aspect=1.0), (256, 256)),
             "close": (Camera(eye=(-1.9, 5.55, 5.25), at=(-2.5, 5.25, 4.1), aspect=1.0), (128, 128)),      # inside the near range of one mesh
             "through": (Camera(eye=(3.2, 1.6, -2.4), at=(-2.5, 1.25, 4.1), fovy=1.0, aspect=2.0), (256, 128))}   # the ray-traced scene in front of a mesh
    scene, sky = bunny_scene(), fixture_sky()
    for name, (cam, (w, h)) in views.items():
        words = raster.overlay_words(cam.view_proj().ravel(), worlds, its, **raster.REFERENCE_LIGHT)
        alone = refrun.run_overlay(v, i, words, mtex, w, h)
        u = Uniforms.from_camera(cam, max_recursion_limit=2).pack()
        traced = refrun.run_reference(scene, u, w, h, sky=sky)
        frame = refrun.run_overlay(v, i, words, mtex, w, h, scene=scene, uniforms70=u, sky=sky)
        out = dict(words=words, uniforms=u, sky=sky, size=np.array([w, h], np.int32),
                   alone_rgba=alone["rgba"], alone_depth24=depth24_of(alone["depth"]),
                   traced_rgba=traced["rgba"], traced_depth24=depth24_of(traced["depth"]),
                   frame_rgba=frame["rgba"], frame_depth24=depth24_of(frame["depth"]))
        if name == "through":      # three frames of App::Render: each ray-tracing draw blends over the FINISHED previous frame (mesh pixels included)
            times = (np.arange(3, dtype=np.float32) / np.float32(60.0)).astype(np.float32)
            three = refrun.run_overlay(v, i, words, mtex, w, h, scene=scene, uniforms70=u, sky=sky, times=times, present=True)
            out.update(times=times, three_rgba=three["rgba"], three_depth24=depth24_of(three["depth"]), three_present=three["present"])
        path = os.path.join(GOLDEN, f"ref_overlay_suzanne_{name}.npz")
        np.savez_compressed(path, **out)
        print("overlay", name, "mesh pixels alone", int((out["alone_depth24"] != 0xFFFFFF).sum()), "in the frame", int((out["frame_depth24"] != out["traced_depth24"]).sum()), os.path.getsize(path))


PRESENT_FRAG = """#version 430
layout(std430, binding = 0) buffer B { readonly vec4 xin[]; };
out vec4 fs_out_col;
void main() { uint i = uint(gl_FragCoord.y) * 64u + uint(gl_FragCoord.x); fs_out_col = xin[i]; }
"""


def present_fixture():
    """ref_present.npz: 4096 colours (ramps beyond [0, 1], values around every tie k + 0.5, NaN and infinities; own probe text) and the
    bytes the window's framebuffer holds after the blit of Framebuffer.cpp:33-35."""
    import subprocess
    import tempfile
    rng = np.random.default_rng(0)
    x = np.zeros((64 * 64, 4), np.float32)
    x[:, 0] = np.linspace(-0.1, 1.1, 4096, dtype=np.float32)
    x[:, 1] = (np.arange(4096) / 4096 * 255 + 0.5).astype(np.float32) / np.float32(255)
    x[:, 2] = rng.uniform(0, 1, 4096).astype(np.float32)
    x[:, 3] = rng.uniform(0, 1, 4096).astype(np.float32)
    x[:8, 2] = [np.nan, np.inf, -np.inf, 0.5, 1.0, 0.0, -0.0, 1e-10]
    x[100:356, 2] = ((np.arange(256) + 0.5) / 255).astype(np.float32)
    with tempfile.TemporaryDirectory() as td:
        pf = os.path.join(td, "p.frag"); open(pf, "w").write(PRESENT_FRAG)
        pv = os.path.join(td, "p.vert")
        open(pv, "w").write("#version 430\nout vec2 vs_out_ndc_coord;\nvoid main(){ vec2 p = vec2(gl_VertexID == 1 ? 3.0 : -1.0, gl_VertexID == 2 ? 3.0 : -1.0); gl_Position = vec4(p, 0.0, 1.0); vs_out_ndc_coord = p; }\n")
        pi = os.path.join(td, "in"); x.tofile(pi)
        subprocess.run([refrun.RUNNER, "--quiet", "--w", "64", "--h", "64", "--vert", pv, "--frag", pf, "--ssbo0", pi, "--noblend", "--present", "--out", os.path.join(td, "o")], check=True, capture_output=True)
        p8 = np.fromfile(os.path.join(td, "o.present.u8"), np.uint8).reshape(-1, 4)
    np.savez_compressed(os.path.join(GOLDEN, "ref_present.npz"), color=x, rgba8=p8)
    print("present", os.path.getsize(os.path.join(GOLDEN, "ref_present.npz")))


def accum8_fixture():
    """ref_accum8.npz: eight progressively blended frames (the blend factors 1/1 .. 1/8), 96 x 64."""
    scene, sky = bunny_scene(), fixture_sky()
    times = (np.arange(8, dtype=np.float32) / np.float32(60.0)).astype(np.float32)
    u = Uniforms.from_camera(Camera(aspect=1.5), max_recursion_limit=3).pack()
    acc = refrun.run_reference(scene, u, 128, 64, sky=sky, times=times)
    np.savez_compressed(os.path.join(GOLDEN, "ref_accum8.npz"), uniforms=u, sky=sky, size=np.array([128, 64], np.int32), times=times, rgba=acc["rgba"])
    print("accum8", os.path.getsize(os.path.join(GOLDEN, "ref_accum8.npz")))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "accum8":
        refrun.build()
        assert refrun.available(), "needs /root/reference and the llvmpipe runner"
        accum8_fixture()
        return
    if len(sys.argv) > 1 and sys.argv[1] in ("synthetic", "ndc"):
        refrun.build()
        assert refrun.available(), "needs /root/reference and the llvmpipe runner"
        if sys.argv[1] == "synthetic":
            synthetic_fixtures()
        else:
            sizes = sys.argv[2:] or ["1280x720", "3840x2160", "7680x4320"]
            if "1280x720" in sizes:
                ndc_fixture(1280, 720, (440, 504), max_recursion_limit=5)
            if "3840x2160" in sizes:
                ndc_fixture(3840, 2160, (1400, 1432), max_recursion_limit=2)
            if "7680x4320" in sizes:
                ndc_fixture(7680, 4320, (2800, 2816), max_recursion_limit=5)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "overlay":
        refrun.build()
        assert refrun.available(), "needs /root/reference and the llvmpipe runner"
        overlay_fixtures()
        present_fixture()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "fullframe":
        refrun.build()
        assert refrun.available(), "needs /root/reference and the llvmpipe runner"
        fullframe_fixture(sys.argv[2:] or None)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "sky":
        refrun.build()
        assert refrun.available(), "needs /root/reference and the llvmpipe runner"
        sky_size_fixture()
        return
    if len(sys.argv) > 2 and sys.argv[1] == "views":       # only the named bunny views
        refrun.build()
        assert refrun.available(), "needs /root/reference and the llvmpipe runner"
        scene, sky = bunny_scene(), fixture_sky()
        for name in sys.argv[2:]:
            v = VIEWS[name]
            w, h = v["size"]
            rec = reference_record(scene, Uniforms.from_camera(v["cam"], **v.get("kw", {})).pack(), w, h, sky)
            np.savez_compressed(os.path.join(GOLDEN, f"ref_{name}.npz"), **rec)
            print(name, w, h, "kinds", np.bincount(rec["kind"].ravel(), minlength=6), "max segments", int(rec["nrays"].max()))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "scenes":      # only the refscene_* set
        refrun.build()
        assert refrun.available(), "needs /root/reference and the llvmpipe runner"
        scene_fixtures()
        return
    refrun.build()
    assert refrun.available(), "needs /root/reference and the llvmpipe runner"
    os.makedirs(GOLDEN, exist_ok=True)
    pos, nrm, faces = oracle.parse_obj(BUNNY_OBJ)
    assert nrm is None
    np.savez_compressed(os.path.join(GOLDEN, "bunny_mesh.npz"), positions=pos, faces=faces.astype(np.uint32))
    scene = bunny_scene()
    sky = fixture_sky()
    for name, v in VIEWS.items():
        w, h = v["size"]
        u = Uniforms.from_camera(v["cam"], **v.get("kw", {})).pack()
        rec = reference_record(scene, u, w, h, sky)
        np.savez_compressed(os.path.join(GOLDEN, f"ref_{name}.npz"), **rec)
        print(name, w, h, "kinds", np.bincount(rec["kind"].ravel(), minlength=6))
    np.savez_compressed(os.path.join(GOLDEN, "ref_math.npz"), **math_fixture(sky))
    # progressive accumulation
    times = np.array([0.0, 1 / 60, 2 / 60, 0.5], np.float32)
    u = Uniforms.from_camera(Camera(aspect=1.0)).pack()
    acc = refrun.run_reference(scene, u, 128, 128, sky=sky, times=times)
    np.savez_compressed(os.path.join(GOLDEN, "ref_accum.npz"), uniforms=u, sky=sky, size=np.array([128, 128], np.int32), times=times, rgba=acc["rgba"])
    # interpolated NDC at the reference's window size
    w, h = 640, 512
    nx, ny = capture_ndc(w, h)
    cx, cy = closed_form_ndc(w, h)
    dx, dy = ulp_delta(nx, cx), ulp_delta(ny, cy)
    assert np.abs(dx).max() < 2 ** 31 and np.abs(dy).max() < 2 ** 31
    assert np.array_equal(apply_ulp_delta(cx, dx).view(np.uint32), nx.view(np.uint32)) and np.array_equal(apply_ulp_delta(cy, dy).view(np.uint32), ny.view(np.uint32))
    u = Uniforms.from_camera(Camera(aspect=w / h)).pack()
    rec = reference_record(scene, u, w, h, sky, rows=(200, 264))
    rec["ndc_dx"], rec["ndc_dy"] = dx.astype(np.int32), dy.astype(np.int32)
    np.savez_compressed(os.path.join(GOLDEN, "ref_ndc_640x512.npz"), **rec)
    print("ndc 640x512: x differs on %.1f%% of pixels, y on %.1f%%" % (100 * (dx != 0).mean(), 100 * (dy != 0).mean()))
    scene_fixtures()
    sky_size_fixture()
    accum8_fixture()
    synthetic_fixtures()
    ndc_fixture(1280, 720, (440, 504), max_recursion_limit=5)
    ndc_fixture(3840, 2160, (1400, 1432), max_recursion_limit=2)
    total = sum(os.path.getsize(os.path.join(GOLDEN, f)) for f in os.listdir(GOLDEN))
    print("fixtures total bytes:", total)


if __name__ == "__main__":
    main()
