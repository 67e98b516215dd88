// Host-side scene build behind rt_mesh_* / rt_octree_* (include/rt_hotpath.h): what the reference's Mesh and Octree
// classes compute (app/core/src/Mesh.cpp:11-101, app/core/src/Octree.cpp:38-85,118-256,
// app/core/include/TriangleBoxIntersection.hpp:153-240), organised for large scenes: nodes live in one pool, are
// split level by level with the independent nodes of a level processed by a thread team, and the flattened arrays
// are written by an explicit-stack walk.  Every floating-point operation is the one the reference performs (same
// operands, same order, no contraction), so the flattened octree is bit-identical to the reference's.
#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

#include "rt_build_internal.h"

extern "C" const char *rt_last_error(void);

namespace rt_host {

static inline void normalize_glm(float v[3]) { const float inv = 1.0f / std::sqrt(dot_xyz(v, v)); v[0] *= inv; v[1] *= inv; v[2] *= inv; }

// ---- octree ------------------------------------------------------------------------------------------------
struct BuildNode {
    float lo[3], hi[3];
    std::vector<Tri> tris;
    int32_t kids[8];     // pool indices, -1 = none
    bool leaf = true;
    uint32_t depth = 0;
    BuildNode() { std::fill(kids, kids + 8, -1); }
};

struct Split {           // result of splitting one node, produced by a worker, merged serially
    bool did_split = false;
    float clo[8][3], chi[8][3];
    std::vector<Tri> child_tris[8];
    std::vector<Tri> kept;
};

static void split_node(const BuildNode &n, const float *verts, const Params &P, Split &out) {
    // Octree.cpp:119-124: stop at the depth limit (depth counted from 1 at the root) or when small enough
    if ((size_t)n.depth + 1 >= P.depth_limit || n.tris.size() <= P.max_per_leaf) return;
    out.did_split = true;
    float center[8][3], halfsz[8][3];
    for (int i = 0; i < 8; i++) {
        child_box(n.lo, n.hi, i, out.clo[i], out.chi[i]);
        box_centre_half(out.clo[i], out.chi[i], center[i], halfsz[i]);
    }
    struct Cand { uint8_t mask, count; Tri tri; };
    std::vector<Cand> cands;
    for (const Tri &t : n.tris) {
        const float *a = verts + 4 * (size_t)t.a, *b = verts + 4 * (size_t)t.b, *c = verts + 4 * (size_t)t.c;
        uint8_t mask = 0, count = 0;
        for (int i = 0; i < 8; i++) if (triangle_overlaps_box(center[i], halfsz[i], a, b, c)) { mask |= (uint8_t)(1u << i); count++; }
        if (count < P.keep_after) { for (int i = 0; i < 8; i++) if (mask & (1u << i)) out.child_tris[i].push_back(t); }   // Octree.cpp:163-166
        else cands.push_back({mask, count, t});
    }
    if (cands.size() > P.max_per_node)   // Octree.cpp:172-181: the library's partial sort, ordering on the overlap count only
        std::partial_sort(cands.begin(), cands.begin() + (std::ptrdiff_t)P.max_per_node, cands.end(), [](const Cand &x, const Cand &y) { return x.count > y.count; });
    const size_t keep = std::min(cands.size(), P.max_per_node);
    for (size_t i = 0; i < keep; i++) out.kept.push_back(cands[i].tri);
    for (size_t i = P.max_per_node; i < cands.size(); i++)
        for (int k = 0; k < 8; k++) if (cands[i].mask & (1u << k)) out.child_tris[k].push_back(cands[i].tri);
}

void root_bounds(const float *verts, const std::vector<Tri> &all, float lo[3], float hi[3]) {
    const float inf = std::numeric_limits<float>::infinity();
    for (int k = 0; k < 3; k++) { lo[k] = inf; hi[k] = -inf; }
    for (const Tri &t : all)   // Octree.cpp:234-246, min/max as glm defines them
        for (uint32_t vi : {t.a, t.b, t.c})
            for (int k = 0; k < 3; k++) {
                const float x = verts[4 * (size_t)vi + k];
                lo[k] = x < lo[k] ? x : lo[k];
                hi[k] = hi[k] < x ? x : hi[k];
            }
}

static int build_octree(const float *verts, size_t n_vert, const std::vector<Tri> &all, const Params &P, rt_octree *out) {
    (void)n_vert;
    std::vector<BuildNode> pool(1);
    float lo[3], hi[3];
    root_bounds(verts, all, lo, hi);
    for (int k = 0; k < 3; k++) { pool[0].lo[k] = lo[k]; pool[0].hi[k] = hi[k]; }
    pool[0].tris = all;
    size_t max_depth = 0;

    std::vector<uint32_t> level = {0};
    unsigned hw = std::thread::hardware_concurrency();
    const unsigned n_threads = std::max(1u, std::min(hw ? hw : 1u, 16u));
    while (!level.empty()) {
        std::vector<Split> splits(level.size());
        std::atomic<size_t> next{0};
        auto work = [&]() {
            for (;;) {
                const size_t j = next.fetch_add(1);
                if (j >= level.size()) break;
                split_node(pool[level[j]], verts, P, splits[j]);
            }
        };
        if (level.size() < 4 * n_threads && pool[level[0]].tris.size() < 4096) work();
        else {
            std::vector<std::thread> team;
            for (unsigned i = 1; i < n_threads; i++) team.emplace_back(work);
            work();
            for (std::thread &th : team) th.join();
        }
        std::vector<uint32_t> next_level;
        for (size_t j = 0; j < level.size(); j++) {
            Split &sp = splits[j];
            const uint32_t ni = level[j];
            const uint32_t depth = pool[ni].depth;
            // Octree.cpp:119-125: m_max_depth is raised for every node below the depth limit
            if ((size_t)depth + 1 < P.depth_limit) max_depth = std::max(max_depth, (size_t)depth + 1);
            if (!sp.did_split) continue;
            pool[ni].leaf = false;
            pool[ni].tris = std::move(sp.kept);
            for (int i = 0; i < 8; i++) {
                BuildNode c;
                for (int k = 0; k < 3; k++) { c.lo[k] = sp.clo[i][k]; c.hi[k] = sp.chi[i][k]; }
                c.tris = std::move(sp.child_tris[i]);
                c.depth = depth + 1;
                const bool recurse = !c.tris.empty();   // Octree.cpp:194-198
                pool.push_back(std::move(c));
                pool[ni].kids[i] = (int32_t)(pool.size() - 1);
                if (recurse) next_level.push_back((uint32_t)(pool.size() - 1));
            }
        }
        level.swap(next_level);
    }

    // Octree.cpp:38-85: pre-order flattening, children in ascending octant order, empty leaves dropped
    std::vector<uint32_t> words;
    std::vector<Tri> refs;
    struct Visit { uint32_t node; size_t parent_slot; };
    std::vector<Visit> stack = {{0u, 0u}};
    while (!stack.empty()) {
        const Visit v = stack.back();
        stack.pop_back();
        const BuildNode &n = pool[v.node];
        if (v.parent_slot != 0) words[v.parent_slot] = (uint32_t)words.size();
        uint32_t mask = 0, count = 0;
        if (!n.leaf) for (int i = 0; i < 8; i++) { const BuildNode &c = pool[n.kids[i]]; if (!c.leaf || !c.tris.empty()) { mask |= 1u << i; count++; } }
        const uint32_t tri_count = (uint32_t)(uint16_t)n.tris.size();   // 16-bit field (Octree.cpp:44)
        words.push_back((tri_count << 16) | (mask << 8) | count);
        words.push_back((uint32_t)refs.size());
        refs.insert(refs.end(), n.tris.begin(), n.tris.end());
        const size_t first = words.size();
        for (uint32_t i = 0; i < count; i++) words.push_back(0u);
        // push in descending octant order so that the lowest octant is flattened first
        size_t slot = first + count;
        for (int i = 7; i >= 0; i--) if (mask & (1u << i)) stack.push_back({(uint32_t)n.kids[i], --slot});
    }

    memset(out, 0, sizeof *out);
    out->n_words = words.size(); out->n_ref = refs.size(); out->max_depth = max_depth;
    out->nodes = (uint32_t *)malloc(std::max<size_t>(1, words.size()) * 4);
    out->tri_refs = (uint32_t *)malloc(std::max<size_t>(1, refs.size()) * 16);
    if (!out->nodes || !out->tri_refs) { free(out->nodes); free(out->tri_refs); memset(out, 0, sizeof *out); return RT_ENOMEM; }
    if (!words.empty()) memcpy(out->nodes, words.data(), words.size() * 4);
    if (!refs.empty()) memcpy(out->tri_refs, refs.data(), refs.size() * 16);
    for (int k = 0; k < 3; k++) { out->bounds_min[k] = lo[k]; out->bounds_max[k] = hi[k]; }
    return RT_OK;
}

// ---- OBJ text -> arrays ------------------------------------------------------------------------------------
struct ObjData { std::vector<float> pos, nrm; std::vector<uint32_t> faces; std::vector<int64_t> face_normal_idx; };

static bool parse_index(const char *&p, size_t n_pos, size_t n_nrm, int64_t &vi, int64_t &ni) {
    char *end;
    long v = strtol(p, &end, 10);
    if (end == p) return false;
    vi = v > 0 ? v - 1 : (int64_t)n_pos + v;
    ni = -1;
    p = end;
    if (*p == '/') {
        p++;
        if (*p != '/') { (void)strtol(p, &end, 10); p = end; }   // texture index, unused
        if (*p == '/') {
            p++;
            long nn = strtol(p, &end, 10);
            if (end != p) { ni = nn > 0 ? nn - 1 : (int64_t)n_nrm + nn; p = end; }
        }
    }
    return true;
}

static int parse_obj(const char *path, ObjData &d, std::string &err) {
    FILE *f = fopen(path, "rb");
    if (!f) { err = std::string("cannot open ") + path + ": " + strerror(errno); return RT_EINVAL; }
    std::string line;
    char buf[4096];
    while (fgets(buf, sizeof buf, f)) {
        line = buf;
        while (!line.empty() && line.back() != '\n' && fgets(buf, sizeof buf, f)) line += buf;
        const char *p = line.c_str();
        while (*p == ' ' || *p == '\t') p++;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            char *e; p += 2;
            for (int k = 0; k < 3; k++) { d.pos.push_back((float)strtod(p, &e)); p = e; }
        } else if (p[0] == 'v' && p[1] == 'n' && (p[2] == ' ' || p[2] == '\t')) {
            char *e; p += 3;
            for (int k = 0; k < 3; k++) { d.nrm.push_back((float)strtod(p, &e)); p = e; }
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            p += 2;
            int64_t vi[4], ni[4];
            int n = 0;
            for (;;) {
                while (*p == ' ' || *p == '\t') p++;
                if (*p == 0 || *p == '\n' || *p == '\r') break;
                int64_t a, b;
                if (!parse_index(p, d.pos.size() / 3, d.nrm.size() / 3, a, b)) { fclose(f); err = "malformed face record"; return RT_ESCENE; }
                if (n < 4) { vi[n] = a; ni[n] = b; }
                n++;
            }
            // The reference reads OBJ files through tinyobjloader with triangulation on (Mesh.cpp:12-14,58-63), so what
            // reaches it are triangles.  Triangles pass through; a quad is cut along its SHORTER diagonal, as that
            // library's 2.0.0 release candidates do (squared lengths in float, (0,1,2)+(0,2,3) when |v0v2|^2 < |v1v3|^2,
            // else (0,1,3)+(1,2,3)); larger polygons (its ear clipping) are not restated and are refused.
            if (n < 3 || n > 4) { fclose(f); err = n < 3 ? "face with fewer than three vertices" : "faces with more than four vertices are not supported"; return RT_ESCENE; }
            int order[6] = {0, 1, 2, 0, 2, 3};
            if (n == 4) {
                for (int k = 0; k < 4; k++)
                    if (vi[k] < 0 || (size_t)vi[k] >= d.pos.size() / 3) { fclose(f); err = "face indexes a vertex that does not exist"; return RT_ESCENE; }
                const float *v0 = &d.pos[3 * (size_t)vi[0]], *v1 = &d.pos[3 * (size_t)vi[1]], *v2 = &d.pos[3 * (size_t)vi[2]], *v3 = &d.pos[3 * (size_t)vi[3]];
                const float e02[3] = {v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2]}, e13[3] = {v3[0] - v1[0], v3[1] - v1[1], v3[2] - v1[2]};
                const float sqr02 = e02[0] * e02[0] + e02[1] * e02[1] + e02[2] * e02[2], sqr13 = e13[0] * e13[0] + e13[1] * e13[1] + e13[2] * e13[2];
                if (!(sqr02 < sqr13)) { const int alt[6] = {0, 1, 3, 1, 2, 3}; for (int k = 0; k < 6; k++) order[k] = alt[k]; }
            }
            for (int k = 0; k < (n == 4 ? 6 : 3); k++) { d.faces.push_back((uint32_t)vi[order[k]]); d.face_normal_idx.push_back(ni[order[k]]); }
        }
    }
    fclose(f);
    return RT_OK;
}

static int mesh_from(const float *pos, const float *file_nrm, size_t n, const uint32_t *faces, size_t m, uint32_t material, const float T[16], rt_mesh *out) {
    memset(out, 0, sizeof *out);
    out->vertices = (float *)malloc(std::max<size_t>(1, n) * 16);
    out->normals = (float *)malloc(std::max<size_t>(1, n) * 16);
    out->triangles = (uint32_t *)malloc(std::max<size_t>(1, m) * 16);
    if (!out->vertices || !out->normals || !out->triangles) { rt_mesh_free(out); return RT_ENOMEM; }
    out->n_vert = n; out->n_tri = m;
    std::vector<float> acc;
    if (!file_nrm) acc.assign(n * 3, 0.0f);
    for (size_t i = 0; i < n; i++) {
        const float x = pos[3 * i], y = pos[3 * i + 1], z = pos[3 * i + 2], w = 0.0f;   // Mesh.cpp:44
        for (int r = 0; r < 4; r++) out->vertices[4 * i + r] = (T[r] * x + T[4 + r] * y) + (T[8 + r] * z + T[12 + r] * w);
        if (file_nrm) { for (int k = 0; k < 3; k++) out->normals[4 * i + k] = file_nrm[3 * i + k]; out->normals[4 * i + 3] = 0.0f; }
        else acc[3 * i] = 1.0f;   // Mesh.cpp:52
    }
    for (size_t f = 0; f < m; f++) {
        const uint32_t a = faces[3 * f], b = faces[3 * f + 1], c = faces[3 * f + 2];
        if (a >= n || b >= n || c >= n) { rt_mesh_free(out); return RT_ESCENE; }
        if (!file_nrm) {   // Mesh.cpp:75-83
            const float *va = out->vertices + 4 * (size_t)a, *vb = out->vertices + 4 * (size_t)b, *vc = out->vertices + 4 * (size_t)c;
            const float e1[3] = {vb[0] - va[0], vb[1] - va[1], vb[2] - va[2]}, e2[3] = {vc[0] - va[0], vc[1] - va[1], vc[2] - va[2]};
            float nr[3];
            cross_glm(e1, e2, nr);
            normalize_glm(nr);
            for (uint32_t vi : {a, b, c}) for (int k = 0; k < 3; k++) acc[3 * (size_t)vi + k] = acc[3 * (size_t)vi + k] + nr[k];
        }
        uint32_t *t = out->triangles + 4 * f;
        t[0] = a; t[1] = b; t[2] = c; t[3] = material;
    }
    if (!file_nrm)
        for (size_t i = 0; i < n; i++) {   // Mesh.cpp:96-99
            float q[3] = {acc[3 * i], acc[3 * i + 1], acc[3 * i + 2]};
            normalize_glm(q);
            for (int k = 0; k < 3; k++) out->normals[4 * i + k] = q[k];
            out->normals[4 * i + 3] = 0.0f;
        }
    return RT_OK;
}

int merge_meshes(const rt_mesh *meshes, size_t n_meshes, float **verts_out, float **norms_out, size_t *n_vert, std::vector<Tri> &all) {
    size_t nv = 0, nt = 0;
    for (size_t i = 0; i < n_meshes; i++) { nv += meshes[i].n_vert; nt += meshes[i].n_tri; }
    if (nv > 0xFFFFFFFFull || nt > 0xFFFFFFFFull) return rt_internal_set_error(RT_ELIMIT, "scene too large for 32-bit indices");
    float *verts = (float *)malloc(std::max<size_t>(1, nv) * 16), *norms = (float *)malloc(std::max<size_t>(1, nv) * 16);
    if (!verts || !norms) { free(verts); free(norms); return rt_internal_set_error(RT_ENOMEM, "out of memory"); }
    all.clear();
    all.reserve(nt);
    size_t base = 0;
    for (size_t i = 0; i < n_meshes; i++) {   // Octree.cpp:221-233
        const rt_mesh &m = meshes[i];
        if (m.n_vert) {
            memcpy(verts + 4 * base, m.vertices, m.n_vert * 16);
            memcpy(norms + 4 * base, m.normals, m.n_vert * 16);
        }
        for (size_t t = 0; t < m.n_tri; t++) {
            const uint32_t *q = m.triangles + 4 * t;
            if (q[0] >= m.n_vert || q[1] >= m.n_vert || q[2] >= m.n_vert) { free(verts); free(norms); return rt_internal_set_error(RT_ESCENE, "triangle indexes a vertex that does not exist"); }
            all.push_back({q[0] + (uint32_t)base, q[1] + (uint32_t)base, q[2] + (uint32_t)base, q[3]});
        }
        base += m.n_vert;
    }
    *verts_out = verts; *norms_out = norms; *n_vert = nv;
    return RT_OK;
}

}  // namespace rt_host

extern "C" {

void rt_mesh_free(rt_mesh *m) {
    if (!m) return;
    free(m->vertices); free(m->normals); free(m->triangles);
    memset(m, 0, sizeof *m);
}

int rt_mesh_from_arrays(const float *positions, const float *file_normals, size_t n, const uint32_t *faces, size_t m,
                        uint32_t material_id, const float transform[16], rt_mesh *out) {
    if (!out || !transform || (n && !positions) || (m && !faces)) return rt_internal_set_error(RT_EINVAL, "null argument");
    if (material_id >= RT_NUM_MATERIALS) return rt_internal_set_error(RT_EINVAL, "material id out of range");
    int rc = rt_host::mesh_from(positions, file_normals, n, faces, m, material_id, transform, out);
    if (rc == RT_ESCENE) return rt_internal_set_error(rc, "face indexes a vertex that does not exist");
    if (rc) return rt_internal_set_error(rc, "out of memory");
    return RT_OK;
}

int rt_mesh_load_obj(const char *path, uint32_t material_id, const float transform[16], rt_mesh *out) {
    if (!path || !out || !transform) return rt_internal_set_error(RT_EINVAL, "null argument");
    rt_host::ObjData d;
    std::string err;
    int rc = rt_host::parse_obj(path, d, err);
    if (rc) return rt_internal_set_error(rc, err.c_str());
    const bool has_normals = !d.nrm.empty();
    if (has_normals && d.nrm.size() != d.pos.size())   // Mesh.cpp:28-33
        return rt_internal_set_error(RT_ESCENE, "obj file contains different amount of vertices as to normals");
    if (has_normals)
        for (size_t i = 0; i < d.faces.size(); i++)      // Mesh.cpp:69-73
            if (d.face_normal_idx[i] != (int64_t)d.faces[i]) return rt_internal_set_error(RT_ESCENE, "different indices for vertices and normals are not supported");
    return rt_mesh_from_arrays(d.pos.data(), has_normals ? d.nrm.data() : nullptr, d.pos.size() / 3, d.faces.data(), d.faces.size() / 3, material_id, transform, out);
}

void rt_octree_free(rt_octree *o) {
    if (!o) return;
    free(o->vertices); free(o->normals); free(o->tri_refs); free(o->nodes);
    memset(o, 0, sizeof *o);
}

int rt_octree_build(const rt_mesh *meshes, size_t n_meshes, size_t depth_limit, size_t max_triangles_per_node,
                    size_t max_triangles_per_leaf, size_t keep_triangles_after_this_many_overlaps, rt_octree *out) {
    if (!out || (n_meshes && !meshes)) return rt_internal_set_error(RT_EINVAL, "null argument");
    if (depth_limit > RT_MAX_TREE_DEPTH) return rt_internal_set_error(RT_ELIMIT, "depth_limit beyond RT_MAX_TREE_DEPTH");
    float *verts = nullptr, *norms = nullptr;
    size_t nv = 0;
    std::vector<rt_host::Tri> all;
    int mrc = rt_host::merge_meshes(meshes, n_meshes, &verts, &norms, &nv, all);
    if (mrc) return mrc;
    rt_host::Params P{depth_limit, max_triangles_per_node, max_triangles_per_leaf, keep_triangles_after_this_many_overlaps};
    int rc = rt_host::build_octree(verts, nv, all, P, out);
    if (rc) { free(verts); free(norms); return rt_internal_set_error(rc, "out of memory"); }
    out->vertices = verts; out->normals = norms; out->n_vert = nv;
    return RT_OK;
}

}  // extern "C"
