// Internal declarations shared by the host API (rt_api.cpp) and the device code (rt_kernels.hip).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <hip/hip_runtime_api.h>
#include "../../include/rt_hotpath.h"

// One octree node as the device sees it: a single 64-byte line holding everything a visit needs -- header,
// triangle range, the record index of its block of children and the node's own box with its midpoint, from which
// child boxes are derived by midpoint split exactly as the reference shader does (ray_tracer.frag:487-500; the midpoint
// (max+min)/2 is computed ahead of time with the same two IEEE operations).  Records are laid out breadth-first; a node
// with children owns a block of eight consecutive records, the child in octant o being record first_child + o (the
// slots of children that do not exist are never read), so a visit turns "octant passes the slab test" into a record
// index without counting mask bits.  Every block starts at a multiple of eight (the root is record 0 of a block of its
// own), so a traversal stack entry names a block by first_child / 8.  A leaf (no children, so no child boxes to derive) carries the record of its first
// triangle in the box fields: lo = v1, mid = e1, hi = e2.  Built from the reference node words (Octree.cpp:44-75) by
// rt_scene_create.
struct alignas(64) RtNodeRec {
    uint32_t info;        // tri_count << 16 | child_mask << 8 | child_count  (reference word 0)
    uint32_t tri_start;   // reference word 1
    uint32_t first_child; // record index of the child in octant 0 (undefined when child_count == 0)
    uint32_t rank;        // position of the node in the order in which the reference shader's traversal meets the nodes of the
                          // whole tree: depth first, a node before its children, children in DESCENDING octant order (frag:456,489-513)
    float lo[3];
    float mid[3];
    float hi[3];
    uint32_t first_id;    // a leaf: RtTriRec::dup_first of its first triangle (the one carried in the box fields); next to hi[2]: one 8-byte load
    uint32_t parent;      // record index of the parent | octant << 27 (RT_NO_PARENT for the root)
    uint32_t pad;
};
#define RT_NO_PARENT 0xFFFFFFFFu
#define RT_PARENT_INDEX(word) ((word) & 0x07FFFFFFu)
#define RT_PARENT_OCTANT(word) (((word) >> 27) & 7u)
static_assert(sizeof(RtNodeRec) == 64, "node record must be one 64-byte line");
#define RT_MAX_SUBFRAMES 8
#define RT_MAX_NODE_RECORDS (1u << 27)   // a traversal stack entry packs the index of a block of eight records into 24 bits

// One triangle reference prepared for Moeller-Trumbore: first vertex and the two edges, computed once per scene with
// the same IEEE subtractions the shader performs per test (frag:256-257), so a test needs three 16-byte loads
// instead of an index load followed by three dependent vertex gathers.
// The same triangle is referenced from every leaf it overlaps (Octree.cpp:146-170).  The references of one triangle --
// equal (i0, i1, i2, material) -- form a RUN in an array sorted by triangle and then by the rank of the node that holds
// the reference: dup_first / dup_pos are the run's first position and this reference's own position in that array, so the
// references of the same triangle that the reference traversal would meet EARLIER than this one are the positions
// [dup_first, dup_pos), and dup_first identifies the triangle.
struct alignas(16) RtTriRec {
    float v1[3]; float e1x;
    float e1y, e1z, e2x, e2y;
    float e2z; uint32_t dup_first, dup_pos, pad;
};
static_assert(sizeof(RtTriRec) == 48, "triangle record is three 16-byte quads");

// What shading a triangle hit reads (frag:541-556), gathered once per scene into one record per reference: the first vertex and
// the two edges -- the barycentric weights are formed from b - a, c - a and p - a (frag:215-217), and the first two ARE the edges,
// by the same IEEE subtractions --, the three vertex normals and the material.  A shaded hit then costs five 16-byte loads from
// one or two cache lines instead of seven loads from seven scattered lines (index record, three vertices, three normals).
struct alignas(16) RtShadeRec {
    float a[3], e1[3], e2[3];
    float n1[3], n2[3], n3[3];
    uint32_t material, pad;
};
static_assert(sizeof(RtShadeRec) == 80, "shading record is five 16-byte quads");

// One entry per distinct triangle for the per-frame search of triangles whose PLANE passes through the camera position (see
// eye_planes_kernel in rt_kernels.hip): unit normal, first vertex, the slot of the triangle's first reference and |e1| + |e2|.
struct alignas(16) RtPlaneRec { float n[3]; float v1x; float v1y, v1z; uint32_t slot; float edges; };
static_assert(sizeof(RtPlaneRec) == 32, "plane record is two 16-byte quads");
#define RT_EYE_CAP 32u       // eye-coplanar triangles a frame can list; more: every primary ray takes the reference's order

struct RtDeviceScene {
    const float *vertices;     // [n_vert][4]  } the caller's arrays as uploaded: only alive while the records below are derived from them
    const float *normals;      // [n_vert][4]  } at scene creation, null afterwards
    const uint32_t *tri_refs;  // [n_ref][4]   }
    const RtNodeRec *nodes;    // [n_nodes]
    const RtTriRec *tris;      // [n_ref]
    const RtShadeRec *shade;   // [n_ref]
    const uint32_t *dup_parent; // [n_ref]: by position in the sorted reference array (see RtTriRec): `parent` word of the node holding that reference
    // Triangles that lie IN a face plane of a node that references them (all three vertices share a coordinate equal to the
    // box bound; the triangles on the faces of the root box, whose bounds are the extreme vertices, are such triangles).  The
    // reference's triangle/box test, in floating point, may or may not list such a triangle in the other nodes whose face it
    // lies in, so a ray can hit it before entering any node that holds it and what the reference's traversal makes of it depends
    // on its visiting order.  A ray that hits one of them is traversed in the reference's own order (see Exact in rt_kernels.hip).
    const RtTriRec *risky_tris; // [n_risky] copies of their records (one per triangle)
    uint32_t n_risky;           // RT_RISKY_TOO_MANY: more than RT_MAX_RISKY, the exact near-first order is not used for this scene
    const RtPlaneRec *planes;   // [n_planes] one per distinct non-degenerate triangle
    uint32_t n_planes;
    const uint8_t *sky;        // [6][sky_h][sky_w][4] or null
    uint32_t n_vert, n_ref, n_nodes;
    int32_t sky_w, sky_h;
    float bounds_min[3], bounds_max[3];
    uint32_t max_depth;        // deepest node level (root = 0)
};

struct RtLaunch {
    rt_uniforms u;
    uint32_t width, height, accum_index, blend;
    uint32_t band_rows, band_parts, band_part, n_local_rows;
    uint32_t compact_rows;
    uint32_t n_cu;                  // compute units of the device (sizes the persistent kernels' grids; 0: 256)
    uint32_t near_first;            // RT_ORDER_*: 0 the reference's child order, 1 nearest first, 2 nearest first with the reference's result (see Exact)
    uint32_t sub_parts, sub_part;   // this launch renders sub-frame sub_part of sub_parts of the caller's band (1, 0: all of it)
    uint32_t hot[4];                // tile rectangle (x0, y0, x1, y1 inclusive, 16x16 tiles of the local rows) where the octree projects to; x1 < x0: none
    // Cost-ordered start of the one-lane-per-pixel kernel (all three null: index order).  order: permutation of the
    // workgroup units to start in (most expensive first), cost: per-unit duration measured by this launch,
    // order_next: where the launch leaves the permutation sorted by that cost for the next frame of the same view.
    const uint32_t *order;
    uint32_t *cost, *order_next;
    const float *ndc_x, *ndc_y;
    float *color, *depth, *hit_t;
    uint32_t *hit_kind, *hit_slot, *n_rays;
    unsigned long long *counters;   // 16 x u64 or null
    // exact near-first order: the frame's eye-coplanar triangles (filled by rt_launch_render: a count and up to RT_EYE_CAP entries of
    // unit normal + slot bits), or null
    const uint32_t *eye_count;
    const float4 *eye_list;
};

// Device workspace of the per-depth wavefront pipeline.  Path segments of one depth are addressed by slot; the
// two copies of the ray planes alternate between "this depth" (index depth & 1) and "next depth".
struct RtWorkspace {
    float4 *pos_seed[2];   // ray origin xyz, RNG seed base
    float4 *dir_pix[2];    // ray direction xyz, target-plane pixel index (uint bits; 0xFFFFFFFF = no pixel)
    float4 *color[2];      // path throughput rgb
    float *t;              // closest distance found so far in this segment
    uint32_t *sphere;      // index of the closest sphere
    uint32_t *tri;         // closest triangle slot | 0x80000000 when a triangle is the closest hit so far
    uint32_t *qt;          // traversal queue: slots of the segments that enter the octree
    uint32_t *live[2];     // per slot range: number of live segments at this / the next depth
    uint32_t *ctr;         // 8 counters: the traversal queues' (QT_*) and, at [4], the number of eye-coplanar triangles of the frame
    float4 *prim;          // per-frame (oc.xyz, c) of the 84 spheres for rays starting at the camera; from entry 96 on: the eye-coplanar triangles
};

// Several consecutive frames of a sequence traced by ONE launch of the one-lane-per-pixel kernel.  Samples of different
// frames are independent; only the blend into the target is ordered.  The launch therefore covers n frames x the
// launch's pixels, writes every sample's tonemapped colour to stage[frame][local pixel], and a second small kernel blends
// the samples of each pixel in frame order.  With that a launch keeps the machine busy even when its share of the frame
// is small (one of 8 ranks' bands), instead of ending on the serial chain of its single longest path n times over.
#define RT_MAX_BATCH 8
// The batched kernel runs one wave per frame: RT_BATCH_BLOCK / 64 frames of an 8x8 pixel tile share a workgroup.  A batch
// that fits one such group is blended inside the kernel; a larger one goes through the staging plane and blend_batch_kernel.
#ifndef RT_BATCH_BLOCK
#define RT_BATCH_BLOCK 256
#endif
static inline uint32_t rt_batch_groups(uint32_t n_frames) { return (n_frames + RT_BATCH_BLOCK / 64u - 1u) / (RT_BATCH_BLOCK / 64u); }
struct RtBatch {
    rt_uniforms u[RT_MAX_BATCH];
    uint32_t accum_index[RT_MAX_BATCH], blend[RT_MAX_BATCH];
    uint32_t n;                 // frames in this launch (2..RT_MAX_BATCH)
    uint32_t stage_stride;      // float4 elements per frame in `stage` (= width x n_local_rows)
    float4 *stage;
};

// Several consecutive frames of a sequence traced by ONE chain of wavefront launches.  The frames share everything but `time`
// (consecutive samples of a still camera, App.cpp:103-140), which only enters ray generation (frag:679): the first kernel of the
// chain generates n x the frame's paths, those of frame f addressed to staging plane f, and every later kernel runs unchanged on
// a queue n times as long -- the sparse end of a launch is shared by n frames.  The planes are blended in frame order afterwards.
#define RT_MAX_STACK 8
struct RtStack {
    uint32_t n;                 // frames (1..RT_MAX_STACK)
    uint32_t stride;            // pixels per staging plane (= width x rows of the rank's share)
    float time[RT_MAX_STACK];
    uint32_t accum_index[RT_MAX_STACK], blend[RT_MAX_STACK];
};

// Optional timing of the dominant kernel (the traversal kernel of the wavefront pipeline, the single kernel of the
// other): pairs of HIP events recorded on the launch stream directly before and after each such launch.
struct RtProbe {
    hipEvent_t *events;   // [2 * capacity]
    uint32_t *launches;   // [capacity]: kernel launches bracketed by each pair
    uint32_t capacity;    // pairs
    uint32_t used;        // pairs recorded so far
};

// rt_kernels.hip
size_t rt_render_units(uint32_t width, uint32_t n_local_rows);   // workgroups of the one-lane-per-pixel kernel
int rt_math_check(int op, unsigned long long out[2]);   // exhaustive check of the kernels' 1/x (0) and sqrt (1): see rt_kernels.hip
size_t rt_pool_bytes();   // size of the persistent path kernel's tile-counter block
// Scene set-up on the device: fills the n_ref triangle records (first vertex and the two edges, frag:256-257) from the
// uploaded vertices and references, then copies every leaf's first triangle record into the leaf's node record.
// Synchronous; returns a hipError_t value.
// Node records from the uploaded node words, breadth-first with eight-slot child blocks; every offset the shader would
// dereference is checked on the way.  RT_OK: `nodes` (hipMalloc'ed, the caller frees) holds n_nodes records.  RT_ESCENE /
// RT_ELIMIT: `error` says what is wrong (1..5: the kinds below, at node word `error_word`).  RT_ENODEVICE: `hip_error`.
#define RT_NODE_ERROR_TOO_DEEP 100u
#define RT_NODE_ERROR_TOO_MANY 101u
#define RT_NODE_ERROR_RECORD_LIMIT 102u
struct RtNodeBuild { RtNodeRec *nodes; uint32_t n_nodes, max_depth, error, error_word; int hip_error; };
#define RT_MAX_RISKY 32u
#define RT_RISKY_TOO_MANY 0xFFFFFFFFu
#define RT_TOP_RECORDS 80u   // records of the three top levels (the root's block of 8, 8 + 64): the part of the tree every entering ray visits
// The plane records of the scene's distinct triangles (hipMalloc'ed, the caller frees); returns a hipError_t value.
int rt_build_plane_records(const RtTriRec *tris, uint32_t n_ref, RtPlaneRec **out, uint32_t *n_out);
int rt_build_node_records(const uint32_t *d_words, size_t n_words, size_t n_ref, const float lo0[3], const float hi0[3], RtNodeBuild *out);
// dup_parent: n_ref words (see RtDeviceScene).
// risky / n_risky: see RtDeviceScene (*risky is hipMalloc'ed when *n_risky is in 1..RT_MAX_RISKY; the caller frees).
int rt_build_scene_records(const float *vertices, const float *normals, const uint32_t *tri_refs, uint32_t n_ref, RtNodeRec *nodes, uint32_t n_nodes, RtTriRec *tris, RtShadeRec *shade,
                           uint32_t *dup_parent, RtTriRec **risky, uint32_t *n_risky);
size_t rt_workspace_slots(uint32_t width, uint32_t n_local_rows, uint32_t stack = 1);    // stack: frames traced by one chain of launches (RtStack)
size_t rt_workspace_ranges(uint32_t width, uint32_t n_local_rows, uint32_t stack = 1);
// blends the staged sample planes of a stack (compact layout of the share) into launch.color, frame after frame
int rt_launch_blend(const RtLaunch &launch, const float4 *stage, const RtStack &stack, void *stream);
// pool != nullptr: persistent path kernel (pool: rt_pool_bytes() of device memory, zeroed by the launch); else workspace == nullptr:
// one-lane-per-pixel kernel; otherwise the per-depth wavefront pipeline.
int rt_launch_render(const RtDeviceScene &scene, const RtLaunch &launch, const RtWorkspace *workspace, uint32_t *pool, void *stream, const char **err, RtProbe *probe = nullptr,
                     const RtBatch *batch = nullptr,    // batch: one-lane-per-pixel pipeline only; launch.u / accum_index / blend are then ignored
                     const RtStack *stack = nullptr);   // stack: wavefront pipeline only; launch.color is then stack->n staging planes, blend off

// rt_api.cpp, for rt_dist.cpp: scene / skybox creation from arrays that are already in device memory (sent by the root rank), the
// library's error slot, a scene's device.
extern "C" {
int rt_scene_create_from(const float *vertices, const float *normals, size_t n_vert, const uint32_t *tri_refs, size_t n_ref,
                         const uint32_t *nodes, size_t n_words, const float bounds_min[3], const float bounds_max[3], int source_is_device, rt_scene **out);
int rt_skybox_set_from(rt_scene *scene, const uint8_t *const faces[6], int w, int h, int source_is_device);
int rt_internal_set_error(int code, const char *msg);
int rt_scene_device(const rt_scene *scene);
int rt_internal_swap_device(int device);   // sets this thread's rt_set_device choice (-1: none) and returns the previous one
}
