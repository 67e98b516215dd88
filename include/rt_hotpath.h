/*
 * rt_hotpath.h -- C ABI of the MI355X ray-tracing pass.
 *
 * Drop-in replacement for the render pass of monkeysoop/raytracer: everything the reference does between
 * m_ray_tracer_shader.Use() and glDrawArrays in App::Render (app/core/src/App.cpp:107-140), the one-time
 * buffer/cubemap uploads that feed it (App.cpp:23-26, Buffer.cpp:3-14, Skybox.cpp:46-57) and the
 * fixed-function accumulate-blend into the RGBA32F target (App.cpp:49,97-105, Framebuffer.cpp:6-12).
 * The work of the fragment shader app/assets/ray_tracer.frag runs as HIP kernels on gfx950.
 *
 * Conventions: plain pointers and sizes; every function returns RT_OK (0) or an RT_E* code and never throws;
 * rt_last_error() gives the message of the calling thread's last failure.  A scene handle may be used by one
 * thread at a time.  There is no CPU fallback: without a usable HIP device every call fails with RT_ENODEVICE.
 */
#ifndef RT_HOTPATH_H
#define RT_HOTPATH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    RT_OK = 0,
    RT_EINVAL = 1,      /* bad argument (null pointer, zero size, unsupported value) */
    RT_ESCENE = 2,      /* scene arrays are inconsistent (index / pointer out of range, tree too deep, ...) */
    RT_ENODEVICE = 3,   /* no HIP device / HIP runtime error */
    RT_ENOMEM = 4,
    RT_ELIMIT = 5       /* size beyond a documented limit */
};

enum { RT_MEM_HOST = 0, RT_MEM_DEVICE = 1 };

/* Primary-hit kinds written to rt_target.hit_kind. */
enum { RT_HIT_MISS = 0, RT_HIT_SPHERE = 1, RT_HIT_TRIANGLE = 2, RT_HIT_CYLINDER = 3, RT_HIT_PORTAL_1 = 4, RT_HIT_PORTAL_2 = 5 };

#define RT_NUM_MATERIALS 7u      /* ray_tracer.frag:95  */
#define RT_MAX_TREE_DEPTH 40u    /* deepest octree the traversal stack is sized for (reference default depth limit: 18) */

typedef struct rt_scene rt_scene;

/* The uniforms of ray_tracer.frag:25-49 as App::Render sets them (App.cpp:120-138); mat4s are column-major as
 * glm::value_ptr yields them.  `width`/`height` (frag:27-28) are unused by the shader and live in rt_frame.
 * octree_min/max_bounds (frag:30-31) belong to the scene.  70 32-bit words. */
typedef struct rt_uniforms {
    float inv_view_proj_mat[16];   /* App.cpp:120 */
    float camera_position[3];      /* App.cpp:121 */
    uint32_t max_recursion_limit;  /* App.cpp:126 (reference: 5) */
    float time;                    /* App.cpp:127 */
    float blur_amount;             /* App.cpp:128 (reference: 1e-5) */
    float portal_position_1[3];    /* App.cpp:129 */
    float portal_direction_1[3];   /* App.cpp:130 */
    float portal_position_2[3];    /* App.cpp:131 */
    float portal_direction_2[3];   /* App.cpp:132 */
    float portal_width;            /* App.cpp:133 */
    float portal_height;           /* App.cpp:134 */
    float portal_1_to_2[16];       /* App.cpp:135 */
    float portal_2_to_1[16];       /* App.cpp:136 */
    float z_near;                  /* App.cpp:137 */
    float z_far;                   /* App.cpp:138 */
} rt_uniforms;

/* One draw of the fullscreen triangle (App.cpp:97-105,140). */
typedef struct rt_frame {
    rt_uniforms u;
    uint32_t width, height;        /* target size in pixels; row 0 is the BOTTOM row (GL window coordinates) */
    uint32_t accum_index;          /* n = m_still_frame_counter at App.cpp:104: dst = src*(1/n) + dst*(1 - 1/n); n >= 1 */
    uint32_t blend;                /* 1: blend as above (GL_BLEND enabled, App.cpp:102); 0: store src (GL_BLEND disabled) */
    /* Screen-space band selection for multi-GPU rendering: rows are cut into strips of `band_rows` rows and strip s
     * is rendered iff s % band_parts == band_part.  band_parts = 1 renders every row. */
    uint32_t band_rows, band_parts, band_part;
    /* 0: target planes are full frames addressed [y][x]; 1: planes hold only this part's rows, packed in band order
     * ([local_row][x] with local_row = (strip / band_parts) * band_rows + y % band_rows) -- the buffer a rank sends
     * when the bands of several GPUs are gathered.  ndc_x / ndc_y stay full-frame. */
    uint32_t compact_rows;
    /* Optional per-pixel NDC of the interpolated varying vs_out_ndc_coord (ray_tracer.vert:5-13), [height][width]
     * floats each, in the memory space of rt_target.memory; NULL = closed form 2*(i+0.5)/size - 1. */
    const float *ndc_x;
    const float *ndc_y;
    void *stream;                  /* hipStream_t to launch on (NULL = default stream) */
} rt_frame;

/* Render target: the colour attachment (RGBA32F, linear, Framebuffer.cpp:6) and optional extra planes.
 * All planes are [height][width] (x4 for colour) addressed by full-frame pixel index (pixels outside the rendered
 * band are left untouched), or, with rt_frame.compact_rows = 1, [rt_band_row_count()][width] holding only the band. */
typedef struct rt_target {
    int32_t memory;                /* RT_MEM_HOST or RT_MEM_DEVICE: where every pointer below (and ndc_x/ndc_y) lives */
    int32_t sync;                  /* 1: return after the frame is complete; 0 (device memory only): return after enqueue */
    float *color;                  /* required; read-modify-write when blend = 1 */
    float *depth;                  /* optional: value the shader writes to gl_FragDepth (frag:671) */
    uint32_t *hit_kind;            /* optional debug planes: primary hit kind (RT_HIT_*), */
    uint32_t *hit_slot;            /*   sphere index or triangle-reference slot (index into tri_refs), */
    float *hit_t;                  /*   hit distance, */
    uint32_t *n_rays;              /*   FindIntersection calls (path segments) of the pixel */
} rt_target;

/* Work counters of the last rt_render on a scene (for roofline accounting). */
typedef struct rt_stats {
    uint64_t rays;                 /* path segments = FindIntersection calls (frag:573) */
    uint64_t node_pops;            /* octree nodes visited (frag:455-461) */
    uint64_t triangle_tests;       /* RayTriangle calls (frag:473) */
    uint64_t child_box_tests;      /* child RayAABB calls (frag:504) */
    uint64_t triangle_hits_shaded; /* segments whose closest hit is a triangle (frag:541-556) */
    uint64_t rays_entering_octree; /* segments passing the root box test (frag:449) */
    uint64_t sphere_tests;         /* RaySphere evaluations actually executed (the shader runs 84 per segment, frag:439-445;
                                      the kernels skip spheres a ray provably cannot be accepted by) */
    uint64_t node_pops_from_lds;   /* of node_pops: visits served from the workgroup's LDS copy of the tree's top three levels */
    uint64_t exact_fallbacks;      /* RT_ORDER_NEAR_FIRST_EXACT: segments re-traversed in the reference's order */
    uint64_t exact_candidate_checks; /* RT_ORDER_NEAR_FIRST_EXACT: boxes of earlier references of the hit triangle tested */
} rt_stats;

/* Library / device ------------------------------------------------------------------------------------------ */
const char *rt_last_error(void);
/* First 16 hex digits of the SHA-256 over the sources of the render kernels and their host side (rt_kernels.hip, rt_internal.h,
 * rt_api.cpp, concatenated in that order) as they were when the library was built: ties a measurement to the code it was taken on. */
const char *rt_build_id(void);
int rt_device_count(int *count);
int rt_set_device(int device);     /* device used by subsequent rt_scene_create calls of this thread */

/* Scene: the four SSBOs the reference uploads at App.cpp:23-26 (bindings 0-3, App.cpp:111-114) and the octree
 * bounds (App.cpp:124-125).  Arrays are copied; the caller keeps ownership (glNamedBufferStorage(...,0) semantics,
 * Buffer.cpp:5).  vertices/normals: [n_vert][4] floats (Octree::m_vertecies/m_normals, Octree.hpp:33-34);
 * tri_refs: [n_ref][4] = (i0,i1,i2,material_id) (m_compressed_triangles, Octree.hpp:36); nodes: the flattened octree
 * words (m_compressed_node_buffer, Octree.hpp:35; format written at Octree.cpp:44-75).  n_ref and n_words may be 0
 * (no mesh).  The node buffer is validated and re-flattened into 64-byte node records for the device; per triangle
 * reference a 48-byte intersection record and an 80-byte shading record (vertices, edges, normals, material) are derived,
 * and the uploaded copies of the four arrays are released: a scene holds about 132 bytes per reference, 32 per distinct
 * triangle (its plane: see RT_ORDER_NEAR_FIRST_EXACT) and 64 per node record (eight records per node with children) in
 * device memory. */
int rt_scene_create(const float *vertices, const float *normals, size_t n_vert,
                    const uint32_t *tri_refs, size_t n_ref,
                    const uint32_t *nodes, size_t n_words,
                    const float bounds_min[3], const float bounds_max[3],
                    rt_scene **out);
/* Cubemap upload (Skybox.cpp:46-57): six RGBA8 faces in the order +x,-x,+y,-y,+z,-z, each [h][w][4] bytes with
 * the first row being the first row of the uploaded surface; sampled LINEAR / CLAMP_TO_EDGE, non-seamless.
 * Without a skybox texture() returns 0 like an unbound texture. */
int rt_skybox_set(rt_scene *scene, const uint8_t *const faces[6], int w, int h);
int rt_render(rt_scene *scene, const rt_frame *frame, const rt_target *target);
/* The reference's progressive accumulation (App::Render called once per displayed frame with an unchanged camera,
 * App.cpp:103-140, blending 1/n at ray_tracer.frag:655-656) as one call: renders frames[0..n_frames) in order into
 * the same target; pixel values are identical to n_frames rt_render calls.  All frames must agree in width, height,
 * band selection, compact_rows, ndc tables and stream; uniforms, accum_index and blend may differ.  With the
 * wavefront and persistent pipelines consecutive frames are traced concurrently on internal streams and blended in
 * frame order (rt_scene_set_frames_in_flight; with sub-frames, their chains are pipelined across frames), which a
 * caller issuing one rt_render per frame on one stream cannot get; a pixel's frames stay ordered either way.
 * Host-memory targets are uploaded once before and downloaded once after the sequence. */
int rt_render_sequence(rt_scene *scene, const rt_frame *frames, size_t n_frames, const rt_target *target);
/* No counterpart in the reference.  For callers that cannot allocate device memory while they render -- code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc: rt_render with the one-kernel pipeline
 * (RT_PIPELINE_MEGAKERNEL, whatever rt_scene_set_pipeline says) whose scratch buffer -- the per-tile cost map and start order
 * of that kernel -- lives in a static device array of the library instead of a scene-owned allocation.  The call allocates
 * and frees nothing, on the device or the host.  Pixels are identical to rt_render's.
 *   - the target must be device memory (RT_MEM_DEVICE); sync = 0 or 1;
 *   - one call covers at most RT_NOALLOC_MAX_TILES tiles of 16x16 pixels of the rows it renders (an 8K frame is 129 600);
 *     more fails with RT_ELIMIT -- render the frame in bands (band_parts) or use rt_render;
 *   - the array (1 MiB) is held by every GPU the library has created a scene on until the process ends, is shared by
 *     ALL scenes and calls on that GPU and keeps the last call's contents (the library clears what must start at zero):
 *     calls that overlap in time on one GPU race -- issue them on one stream, or order them yourself;
 *   - the first call of a process on a device loads the kernels' code: make one call before starting a capture;
 *   - a call made while its stream is being captured does not touch the array at all (its tiles start in index order instead of
 *     the measured-cost order, pixels unchanged), so a replayed graph cannot meet an order another call left there; it also
 *     returns without waiting whatever `sync` says, since the work only runs when the graph is launched. */
#define RT_NOALLOC_MAX_TILES (1u << 17)
int rt_render_noalloc(rt_scene *scene, const rt_frame *frame, const rt_target *target);
/* Number of rows a compact band buffer holds for this frame's band selection (whole strips; the last strip may
 * extend past `height`, those rows are never written). */
uint32_t rt_band_row_count(const rt_frame *frame);
int rt_scene_stats(rt_scene *scene, rt_stats *out);   /* counters of the last rt_render with sync = 1 or after a stream sync */
/* Enables (1) / disables (0) work counters for subsequent renders of this scene (default off: counting costs time). */
int rt_scene_set_counting(rt_scene *scene, int enabled);
/* Measurement aid (no counterpart in the reference): while enabled, every launch of the dominant kernel -- the octree
 * traversal kernel of the wavefront pipeline, the single kernel of the megakernel pipeline -- is bracketed by HIP
 * events on the stream it is launched on (at most 4096 launches per sub-frame stream are kept between reads).
 * rt_scene_kernel_time synchronises the device, returns the summed duration and the number of launches since the last
 * read, and clears the record. */
int rt_scene_set_kernel_timing(rt_scene *scene, int enabled);
int rt_scene_kernel_time(rt_scene *scene, double *total_ms, uint32_t *launches);
/* Selects how a frame is executed (results are bit-identical):
 *   RT_PIPELINE_AUTO (default)  the scene times the three pipelines below on its own frames (a few frames after a change
 *       of frame size, recursion limit or band split, and again now and then) and uses the fastest: which one that is
 *       depends on how much of the work is octree traversal;
 *   RT_PIPELINE_PERSISTENT  one kernel of resident waves; every lane carries one path at a time, lanes
 *       alternate between octree traversal and a batched "segment block" (hit assembly, scattering, next pixel,
 *       sphere tests), and free lanes take new pixels from global tile counters;
 *   RT_PIPELINE_WAVEFRONT  one pass per path depth: sphere tests, a persistent octree-traversal kernel fed from a
 *       queue, shading with compaction of the surviving paths into the next depth's queue;
 *   RT_PIPELINE_MEGAKERNEL  a pixel's whole path in one lane of one kernel: fastest when most rays miss the octree. */
enum { RT_PIPELINE_MEGAKERNEL = 0, RT_PIPELINE_WAVEFRONT = 1, RT_PIPELINE_PERSISTENT = 2, RT_PIPELINE_AUTO = 3 };
int rt_scene_set_pipeline(rt_scene *scene, int pipeline);
/* The pipeline the next frame would run with and, under RT_PIPELINE_AUTO, the per-frame milliseconds measured so far
 * for RT_PIPELINE_MEGAKERNEL / WAVEFRONT / PERSISTENT (index = enum value; negative: not measured yet). */
int rt_scene_pipeline_state(rt_scene *scene, int *selected, double ms_per_frame[3]);
/* Persistent and wavefront pipelines: a frame is rendered as up to n (1..8) interleaved sub-frames (every n-th 16-row
 * strip) whose kernels run on separate internal streams, forked from and joined to rt_frame.stream, so that the sparse
 * end of one sub-frame's kernels overlaps dense work of another.  A sub-frame is never smaller than 2^19 pixels.
 * n = 0 (the default) leaves the number to the library: 3 for a single frame, 1 where a sequence can keep frames in
 * flight instead (see below).  Results do not depend on n. */
int rt_scene_set_subframes(rt_scene *scene, int n);
/* Persistent and wavefront pipelines inside rt_render_sequence: n (1..8) consecutive frames of the sequence are traced
 * concurrently, each on an internal stream and in a workspace of its own (about 112 bytes per pixel of the share for the
 * wavefront pipeline) into a staging plane (16 bytes per pixel and frame in flight); their samples are blended into the
 * target strictly in frame order, so the target holds exactly what frame-after-frame rendering leaves in it, and the other
 * planes hold the last frame's values.  Unlike sub-frames this overlaps kernels without shrinking them and also works
 * when the share is small (one of several ranks' bands, a 720p frame).  n = 0 (the default) leaves the number to the
 * library: 3 unless sub-frames were asked for.  Applies to sequences whose every frame has max_recursion_limit >= 1 and
 * whose target is compact or covers the whole frame; sub-frames times frames in flight is at most 8.
 * No counterpart in the reference, which submits one draw per frame (App.cpp:95-111). */
int rt_scene_set_frames_in_flight(rt_scene *scene, int n);
/* Wavefront pipeline inside rt_render_sequence, together with frames in flight: up to n (1..8) consecutive frames that differ
 * in rt_uniforms.time only -- the samples of a still camera (App.cpp:103-140) -- are traced by ONE chain of launches n times
 * as large: time only enters ray generation (ray_tracer.frag:679), so the first kernel generates the paths of all n frames
 * and the others run unchanged on a queue n times as long, whose sparse end the n frames share.  Samples go to a staging
 * plane per frame and are blended in frame order as above; results are identical.  n = 0 (the default) leaves the number
 * to the library (8, at most 2^25 paths per chain: 4 at 4K; a chain's workspace is about 112 bytes per path, 3.7 GB at that
 * cap, and every frame in flight has one).  Not used when the target has planes besides colour.
 * No counterpart in the reference. */
int rt_scene_set_frames_per_launch(rt_scene *scene, int n);
/* One-kernel-per-frame pipeline inside rt_render_sequence: when a launch covers only a small part of a 4K frame (one
 * rank's bands, a small window), up to max_frames (2..8) consecutive frames are traced by one launch -- their samples
 * are independent -- and blended into the target in frame order by a second small kernel; the pixels are the same.
 * 0 (default): chosen from the size of the launch; 1: never. */
int rt_scene_set_sequence_batch(rt_scene *scene, int max_frames);
/* Order in which the children of an octree node are visited (no counterpart in the reference, whose order is fixed).
 * RT_ORDER_AUTO (default): every output bit-identical to the shader's, by whichever of RT_ORDER_REFERENCE and
 * RT_ORDER_NEAR_FIRST_EXACT the scene measures to be faster for the pipeline at hand -- it times them on its own frames like
 * it times the pipelines (rt_scene_selection_state).  While work counters are on (rt_scene_set_counting) the automatic order is
 * the reference's, whose work the counters are defined as.
 * RT_ORDER_REFERENCE: the shader's own -- passing children in descending octant order whatever the ray's
 * direction (frag:489-513) -- with which every output is bit-identical to the shader's by construction.
 * RT_ORDER_NEAR_FIRST: the children on the ray's near side first.  Closer hits are found earlier, later box tests reject
 * more (5 % fewer node visits on the reference's scene, 60 % on a scene of 1.6 M triangles) and frames get faster.  The
 * closest hit is searched over a different set of nodes, so a result can differ from the shader's where floating-point
 * box and triangle tests disagree about a hit on a box boundary, and where two references or two triangles give exactly
 * the same distance the other one may be reported (same distance, the same triangle when the references are duplicates).
 * This mode is held to the tolerance bar -- colour within 1/255 on at least 99.9 % of the pixels -- not to bit identity.
 * RT_ORDER_NEAR_FIRST_EXACT: the children nearest first AND the reference's result: among the references of the hit triangle
 * the one the shader's traversal meets first is reported (every node carries its rank in the shader's visiting order), and a
 * ray for which that cannot be decided locally is traversed again in the shader's own order.  That is the case when an earlier
 * reference of the triangle lies in a box the ray enters behind the hit point; when another triangle is hit at the same
 * distance or within 2^-18 of it; when a box is entered within 2^-18 of the closest distance known at that moment, so that
 * rounding could have turned the pruning decision the other way; when the ray hits a triangle lying in a face plane of the
 * boxes around it; and when a primary ray lies in the plane of a triangle that also contains the camera position (the
 * triangle test's determinant is rounding noise then and the shader may report the triangle at any distance, depending only on
 * when it tests it: the triangles concerned are found per frame).  What this does not cover by construction is a hit from a
 * box that the ray enters MORE than that band behind the hit point while the box containing the hit point does not list the
 * triangle (a ray grazing a box face along which the float triangle/box test of the build and the ray's slab test disagree),
 * and the same noise on a secondary ray whose origin and direction both lie in another triangle's plane: see the kernels'
 * comment.  Every
 * bit-exactness test of this repository also runs in this mode.  Wavefront and persistent pipelines; the one-kernel pipeline
 * runs the reference's order in this mode (its waves' rays are coherent and most of them never enter the octree: near first
 * saves it 5 % of its node visits on the reference's scene, and on deep scenes the wave-wide pipelines are 2-3x faster anyway). */
#define RT_ORDER_REFERENCE 0
#define RT_ORDER_NEAR_FIRST 1
#define RT_ORDER_NEAR_FIRST_EXACT 2
#define RT_ORDER_AUTO 3
int rt_scene_set_traversal_order(rt_scene *scene, int order);
/* What automatic selection has measured and would run next: pipeline (RT_PIPELINE_*), child order (RT_ORDER_REFERENCE or
 * RT_ORDER_NEAR_FIRST_EXACT under RT_ORDER_AUTO, else the order that was set, as far as that pipeline has it) and the per-frame
 * milliseconds of the five candidates: one-kernel; wavefront and persistent in the reference's order; wavefront and persistent in
 * the exact near-first order (negative: not measured, or not a candidate under the scene's settings). */
int rt_scene_selection_state(rt_scene *scene, int *pipeline, int *order, double ms_per_frame[5]);
/* Pipeline and child order the LAST render call of this scene actually ran with (RT_EINVAL before the first one): the
 * order reads RT_ORDER_REFERENCE where RT_ORDER_NEAR_FIRST_EXACT was asked for and the reference's order ran instead (the
 * one-kernel pipeline; a scene with more than 32 face-coplanar triangles). */
int rt_scene_last_mode(rt_scene *scene, int *pipeline, int *order);
int rt_scene_destroy(rt_scene *scene);

/* Host-side scene build: mirrors of the reference's Mesh and Octree classes --------------------------------- */

/* Mesh::Mesh (Mesh.cpp:11-101) on an OBJ file with triangle or quad faces (quads are cut along the shorter diagonal, the
 * rule of the OBJ library the reference reads files with; larger polygons are refused): positions transformed by the column-major
 * 4x4 `transform` with w = 0 (Mesh.cpp:44), normals from the file or accumulated face normals seeded with (1,0,0)
 * (Mesh.cpp:52,79-83,96-99), triangles (i0,i1,i2,material_id).  Outputs are malloc'ed; free with rt_mesh_free. */
typedef struct rt_mesh {
    float *vertices;   /* [n_vert][4] */
    float *normals;    /* [n_vert][4] */
    uint32_t *triangles; /* [n_tri][4] */
    size_t n_vert, n_tri;
} rt_mesh;
int rt_mesh_load_obj(const char *path, uint32_t material_id, const float transform[16], rt_mesh *out);
/* Same from in-memory OBJ data: positions [n][3], optional file normals [n][3] (or NULL), faces [m][3] zero-based. */
int rt_mesh_from_arrays(const float *positions, const float *file_normals, size_t n, const uint32_t *faces, size_t m,
                        uint32_t material_id, const float transform[16], rt_mesh *out);
void rt_mesh_free(rt_mesh *mesh);

/* Octree::Octree (Octree.cpp:203-256): merges the meshes, computes the root box, subdivides
 * (Octree.cpp:118-201, triangle/box test TriangleBoxIntersection.hpp:153-240) and flattens (Octree.cpp:38-85).
 * Parameters as the reference constructor (Octree.hpp:27): depth_limit, max_triangles_per_node,
 * max_triangles_per_leaf, keep_triangles_after_this_many_overlaps (reference scene: 18, 10, 6, 6, App.cpp:22). */
typedef struct rt_octree {
    float *vertices;       /* m_vertecies  [n_vert][4] */
    float *normals;        /* m_normals    [n_vert][4] */
    uint32_t *tri_refs;    /* m_compressed_triangles [n_ref][4] */
    uint32_t *nodes;       /* m_compressed_node_buffer [n_words] */
    size_t n_vert, n_ref, n_words;
    float bounds_min[3], bounds_max[3];   /* GetMinBounds / GetMaxBounds */
    size_t max_depth;
} rt_octree;
int rt_octree_build(const rt_mesh *meshes, size_t n_meshes, size_t depth_limit, size_t max_triangles_per_node,
                    size_t max_triangles_per_leaf, size_t keep_triangles_after_this_many_overlaps, rt_octree *out);
/* The same construction with Subdivide's triangle/box overlap tests and the distribution of the triangle references to
 * the children (Octree.cpp:146-170,182-192; TriangleBoxIntersection.hpp:153-240) run on HIP device `device`, level by
 * level; only the "keep" selection (Octree.cpp:172-181), a library sort of a few records per node, stays on the host.  Output
 * arrays are host memory, bit-identical to rt_octree_build's.  Fails with RT_ENODEVICE without a GPU. */
int rt_octree_build_device(const rt_mesh *meshes, size_t n_meshes, size_t depth_limit, size_t max_triangles_per_node,
                           size_t max_triangles_per_leaf, size_t keep_triangles_after_this_many_overlaps, int device,
                           rt_octree *out);
void rt_octree_free(rt_octree *octree);
/* Octree::Octree (App.cpp:22) and the four buffer uploads that follow it (App.cpp:23-26) as ONE call that never leaves the device:
 * the octree is built by the device builder and the scene is created from the flattened arrays where they are, without the copy
 * to host memory and back (for the 1.28 M-triangle test scene that copy is 40 of the builder's 100 ms, the upload 8 of the scene's
 * 18).  The scene renders exactly what rt_scene_create renders from rt_octree_build's arrays.  bounds_min / bounds_max (may be
 * NULL) receive GetMinBounds / GetMaxBounds for the uniforms (App.cpp:124-125). */
int rt_scene_create_from_meshes(const rt_mesh *meshes, size_t n_meshes, size_t depth_limit, size_t max_triangles_per_node,
                                size_t max_triangles_per_leaf, size_t keep_triangles_after_this_many_overlaps, int device,
                                float bounds_min[3], float bounds_max[3], rt_scene **out);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOTPATH_H */
