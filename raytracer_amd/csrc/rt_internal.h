// Internal declarations shared by the host API (rt_api.cpp) and the device code (rt_kernels.hip).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "../../include/rt_hotpath.h"

// One octree node as the device sees it: a single 64-byte line holding everything a visit needs -- header,
// triangle range, the eight child slots (indexed by octant, RT_NO_CHILD where the reference's child mask has no
// bit) and the node's own box, from which child boxes are derived by midpoint split exactly as the reference
// shader does (ray_tracer.frag:487-500).  Built from the reference node words (Octree.cpp:44-75) by rt_scene_create.
struct alignas(64) RtNodeRec {
    uint32_t info;        // tri_count << 16 | child_mask << 8 | child_count  (reference word 0)
    uint32_t tri_start;   // reference word 1
    uint32_t child[8];    // record index of the child in octant i
    float lo[3];
    float hi[3];
};
static_assert(sizeof(RtNodeRec) == 64, "node record must be one 64-byte line");
#define RT_NO_CHILD 0xFFFFFFFFu

// One triangle reference prepared for Moeller-Trumbore: first vertex and the two edges, computed on the host with
// the same IEEE subtractions the shader performs per test (frag:256-257), so a test needs three 16-byte loads
// instead of an index load followed by three dependent vertex gathers.
struct alignas(16) RtTriRec {
    float v1[3]; float e1x;
    float e1y, e1z, e2x, e2y;
    float e2z; uint32_t pad[3];
};
static_assert(sizeof(RtTriRec) == 48, "triangle record is three 16-byte quads");

struct RtDeviceScene {
    const float *vertices;     // [n_vert][4]
    const float *normals;      // [n_vert][4]
    const uint32_t *tri_refs;  // [n_ref][4]
    const RtNodeRec *nodes;    // [n_nodes]
    const RtTriRec *tris;      // [n_ref]
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
    const float *ndc_x, *ndc_y;
    float *color, *depth, *hit_t;
    uint32_t *hit_kind, *hit_slot, *n_rays;
    unsigned long long *counters;   // 8 x u64 or null
};

// rt_kernels.hip
int rt_launch_render(const RtDeviceScene &scene, const RtLaunch &launch, void *stream, const char **err);
