// Shared by the host and the device octree builders (rt_host_build.cpp, rt_device_build.hip): the arithmetic of the
// reference's Subdivide step, written once so that both execute the same IEEE operations in the same order
// (app/core/src/Octree.cpp:15-21,128-142, app/core/include/TriangleBoxIntersection.hpp:153-240).  Must be compiled
// without floating-point contraction and with denormals preserved.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/rt_hotpath.h"

#if defined(__HIPCC__)
#define RT_HD __host__ __device__
#else
#define RT_HD
#endif

namespace rt_host {

struct Tri { uint32_t a, b, c, material; };
struct Params { size_t depth_limit, max_per_node, max_per_leaf, keep_after; };

// ---- arithmetic of the glm calls the reference makes -------------------------------------------------------
RT_HD static inline float dot_xyz(const float p[3], const float q[3]) { const float x = p[0] * q[0], y = p[1] * q[1], z = p[2] * q[2]; return (x + y) + z; }
RT_HD static inline void cross_glm(const float p[3], const float q[3], float out[3]) {
    out[0] = p[1] * q[2] - q[1] * p[2];
    out[1] = p[2] * q[0] - q[2] * p[0];
    out[2] = p[0] * q[1] - q[0] * p[1];
}

// ---- triangle / box separating-axis test (TriangleBoxIntersection.hpp:153-240) -----------------------------
// Axis cross(edge, unit axis k): the two non-k coordinates (i, j) of the edge weight the (i, j) coordinates of two
// of the translated vertices; which two vertices are projected is fixed per (edge, axis) in the reference:
// edge 0: X01, Y02, Z12; edge 1: X01, Y02, Z0 (vertices 0, 1); edge 2: X2 (vertices 0, 1), Y1 (0, 1), Z12.
RT_HD static inline int plan_first(int edge, int axis) { return (edge == 0 && axis == 2) || (edge == 2 && axis == 2) ? 1 : 0; }
RT_HD static inline int plan_second(int edge, int axis) { return (edge == 1 && axis == 2) || (edge == 2 && axis < 2) ? 1 : 2; }
RT_HD static inline float project(int axis, const float e[3], const float v[3]) {
    switch (axis) {
    case 0: return e[2] * v[1] - e[1] * v[2];     // a*v.y - b*v.z with (a,b) = (e.z, e.y)
    case 1: return -e[2] * v[0] + e[0] * v[2];    // -a*v.x + b*v.z with (a,b) = (e.z, e.x)
    default: return e[1] * v[0] - e[0] * v[1];    // a*v.x - b*v.y with (a,b) = (e.y, e.x)
    }
}
RT_HD static inline float radius(int axis, const float fe[3], const float h[3]) {
    switch (axis) {
    case 0: return fe[2] * h[1] + fe[1] * h[2];
    case 1: return fe[2] * h[0] + fe[0] * h[2];
    default: return fe[1] * h[0] + fe[0] * h[1];
    }
}
RT_HD static inline bool triangle_overlaps_box(const float center[3], const float h[3], const float t0[3], const float t1[3], const float t2[3]) {
    float v[3][3], e[3][3];
    for (int k = 0; k < 3; k++) { v[0][k] = t0[k] - center[k]; v[1][k] = t1[k] - center[k]; v[2][k] = t2[k] - center[k]; }
    for (int k = 0; k < 3; k++) { e[0][k] = v[1][k] - v[0][k]; e[1][k] = v[2][k] - v[1][k]; e[2][k] = v[0][k] - v[2][k]; }
    for (int ed = 0; ed < 3; ed++) {
        const float fe[3] = {__builtin_fabsf(e[ed][0]), __builtin_fabsf(e[ed][1]), __builtin_fabsf(e[ed][2])};
        for (int ax = 0; ax < 3; ax++) {
            const float pa = project(ax, e[ed], v[plan_first(ed, ax)]);
            const float pb = project(ax, e[ed], v[plan_second(ed, ax)]);
            const float lo = pa < pb ? pa : pb, hi = pa < pb ? pb : pa;
            const float rad = radius(ax, fe, h);
            if (lo > rad || hi < -rad) return false;
        }
    }
    for (int k = 0; k < 3; k++) {
        float lo = v[0][k], hi = v[0][k];
        if (v[1][k] < lo) lo = v[1][k];
        if (v[1][k] > hi) hi = v[1][k];
        if (v[2][k] < lo) lo = v[2][k];
        if (v[2][k] > hi) hi = v[2][k];
        if (lo > h[k] || hi < -h[k]) return false;
    }
    float nrm[3];
    cross_glm(e[0], e[1], nrm);
    float near_[3], far_[3];
    for (int k = 0; k < 3; k++) {
        if (nrm[k] > 0.0f) { near_[k] = -h[k] - v[0][k]; far_[k] = h[k] - v[0][k]; }
        else { near_[k] = h[k] - v[0][k]; far_[k] = -h[k] - v[0][k]; }
    }
    if (dot_xyz(nrm, near_) > 0.0f) return false;
    return dot_xyz(nrm, far_) >= 0.0f;
}

// Child octant i of the box [lo, hi]: bit k of i selects the upper half along axis k (Octree.cpp:130-142); the centre and
// half size handed to the overlap test are derived from the child's corners (Octree.cpp:19-20).
RT_HD static inline void child_box(const float lo[3], const float hi[3], int i, float clo[3], float chi[3]) {
    for (int k = 0; k < 3; k++) {
        const float mid = (lo[k] + hi[k]) / 2.0f;
        const bool upper = (i >> k) & 1;
        clo[k] = upper ? mid : lo[k];
        chi[k] = upper ? hi[k] : mid;
    }
}
RT_HD static inline void box_centre_half(const float clo[3], const float chi[3], float centre[3], float half[3]) {
    for (int k = 0; k < 3; k++) { centre[k] = (clo[k] + chi[k]) / 2.0f; half[k] = (chi[k] - clo[k]) / 2.0f; }
}

// Merges the meshes into one vertex / normal array and one triangle list with rebased indices (Octree.cpp:221-233);
// verts/norms are malloc'ed [n_vert][4] arrays owned by the caller afterwards.  Returns an RT_* code (error text set).
int merge_meshes(const rt_mesh *meshes, size_t n_meshes, float **verts, float **norms, size_t *n_vert, std::vector<Tri> &all);
// Root bounds: component-wise min / max over the triangle vertices as glm defines them (Octree.cpp:234-246).
void root_bounds(const float *verts, const std::vector<Tri> &all, float lo[3], float hi[3]);

}  // namespace rt_host

extern "C" int rt_internal_set_error(int code, const char *msg);
