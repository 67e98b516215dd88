// Device-side scene set-up behind rt_scene_create: from the reference's arrays as uploaded (vertices, triangle references,
// node words: app/core/src/Octree.cpp:44-75, read by the shader at app/assets/ray_tracer.frag:460-500) to the records the
// render kernels traverse.  Compiled with the render kernels' arithmetic mode (no contraction, denormals flushed like the
// shader's), because the child boxes and triangle edges computed here are the shader's own operations done ahead of time.
//
//   node records      the words are walked breadth-first, one tree level per pass: check_nodes_kernel validates every
//                     offset the shader would dereference and says which nodes have children; a prefix sum gives each of
//                     those a block of eight consecutive records in the next level (child in octant o = first_child + o,
//                     slots of missing children stay zero); write_nodes_kernel writes the level's records and the next
//                     level's (word offset, box) list.  The host only sees two counters per level.
//   triangle records  first vertex and the two edges of every reference (frag:256-257, 469-471)
//   leaf triangles    a leaf has no child boxes to derive, so its box fields carry its first triangle record instead
//                     (nine floats either way): visiting a leaf with one triangle takes one memory round trip, not two
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <vector>

#include "rt_internal.h"

namespace {

constexpr uint32_t HOLE = 0xFFFFFFFFu;
constexpr int BLOCK = 256;
enum NodeError : uint32_t { E_NONE = 0, E_PAST_END, E_COUNT_MASK, E_COUNT_BITS, E_POINTERS_PAST_END, E_TRIANGLE_RANGE };

struct Entry { uint32_t word; float lo[3], hi[3]; };            // a node of the level being written (word == HOLE: an empty slot)
struct Tally { uint32_t blocks, children; };                   // nodes with children / child pointers before an entry
struct AddTally { __host__ __device__ Tally operator()(const Tally &a, const Tally &b) const { return Tally{a.blocks + b.blocks, a.children + b.children}; } };

__global__ __launch_bounds__(BLOCK) void check_nodes_kernel(const uint32_t *__restrict__ words, uint32_t n_words, uint32_t n_ref, const Entry *__restrict__ level, uint32_t n,
                                                            Tally *__restrict__ tally, uint32_t *__restrict__ error) {
    const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
    if (j > n) return;
    Tally t{0u, 0u};
    if (j < n && level[j].word != HOLE) {
        const uint32_t w = level[j].word;
        uint32_t bad = E_NONE;
        if ((unsigned long long)w + 2ull > n_words) bad = E_PAST_END;
        else {
            const uint32_t info = words[w], tri_start = words[w + 1];
            const uint32_t count = info & 0xFu, mask = (info >> 8) & 0xFFu, tri_count = info >> 16;
            if ((uint32_t)__popc(mask) != count) bad = E_COUNT_MASK;
            else if ((info & 0xF0u) != 0u) bad = E_COUNT_BITS;
            else if ((unsigned long long)w + 2ull + count > n_words) bad = E_POINTERS_PAST_END;
            else if ((unsigned long long)tri_start + tri_count > n_ref) bad = E_TRIANGLE_RANGE;
            else { t.blocks = count ? 1u : 0u; t.children = count; }
        }
        if (bad != E_NONE && atomicCAS(&error[0], 0u, bad) == 0u) error[1] = w;
    }
    tally[j] = t;       // entry n is the zero that makes the exclusive sum return the totals
}

__global__ __launch_bounds__(BLOCK) void write_nodes_kernel(const uint32_t *__restrict__ words, const Entry *__restrict__ level, uint32_t n, const Tally *__restrict__ before,
                                                            uint32_t next_base, RtNodeRec *__restrict__ recs, Entry *__restrict__ next) {
    const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= n) return;
    RtNodeRec r{};
    const Entry e = level[j];
    if (e.word != HOLE) {
        const uint32_t info = words[e.word], count = info & 0xFu, mask = (info >> 8) & 0xFFu;
        r.info = info;
        r.tri_start = words[e.word + 1];
        for (int k = 0; k < 3; k++) { r.lo[k] = e.lo[k]; r.hi[k] = e.hi[k]; r.mid[k] = (e.hi[k] + e.lo[k]) / 2.0f; }   // frag:487
        if (count) {
            const uint32_t slot0 = 8u * before[j].blocks;
            r.first_child = next_base + slot0;
            uint32_t ci = 0;
            for (uint32_t o = 0; o < 8u; o++) {
                Entry c;
                c.word = HOLE;
                for (int k = 0; k < 3; k++) { c.lo[k] = 0.0f; c.hi[k] = 0.0f; }
                if (mask & (1u << o)) {
                    c.word = words[e.word + 2u + ci++];
                    for (int k = 0; k < 3; k++) {   // frag:491-500
                        const bool upper = (o >> k) & 1u;
                        c.lo[k] = upper ? r.mid[k] : r.lo[k];
                        c.hi[k] = upper ? r.hi[k] : r.mid[k];
                    }
                }
                next[slot0 + o] = c;
            }
        }
    }
    recs[j] = r;
}

// Subtree sizes of one level (children's sizes are final: levels are processed deepest first).  An empty slot has size 0.
__global__ __launch_bounds__(BLOCK) void subtree_size_kernel(const RtNodeRec *__restrict__ recs, uint32_t first, uint32_t n, uint32_t *__restrict__ size, bool root_level) {
    const uint32_t j = first + blockIdx.x * BLOCK + threadIdx.x;
    if (j >= first + n) return;
    const RtNodeRec &r = recs[j];
    const uint32_t mask = (r.info >> 8) & 0xFFu;
    // a slot of a child block that no child occupies is all zero; a real node without triangles and children is too, but a
    // size of 1 or 0 for it changes no comparison between ranks of nodes that hold triangles
    uint32_t s = (r.info == 0u && !(root_level && j == 0u)) ? 0u : 1u;
    for (uint32_t o = 0; o < 8u; o++) if (mask & (1u << o)) s += size[r.first_child + o];
    size[j] = s;
}
// Ranks of the children of one level's nodes: the reference pops the children of a node in descending octant order, each
// with its whole subtree (frag:456,489-513).
__global__ __launch_bounds__(BLOCK) void child_rank_kernel(RtNodeRec *__restrict__ recs, uint32_t first, uint32_t n, const uint32_t *__restrict__ size) {
    const uint32_t j = first + blockIdx.x * BLOCK + threadIdx.x;
    if (j >= first + n) return;
    const uint32_t mask = (recs[j].info >> 8) & 0xFFu, fc = recs[j].first_child;
    uint32_t run = recs[j].rank + 1u;
    for (int o = 7; o >= 0; o--) {
        if (!(mask & (1u << o))) continue;
        recs[fc + o].rank = run;
        recs[fc + o].parent = j | ((uint32_t)o << 27);
        run += size[fc + o];
    }
}
// Which node holds each triangle reference (its rank and parent word); references no node holds keep RT_NO_PARENT.
__global__ __launch_bounds__(BLOCK) void slot_owner_kernel(const RtNodeRec *__restrict__ recs, uint32_t n_nodes, uint32_t n_ref, uint32_t *__restrict__ slot_rank, uint32_t *__restrict__ slot_parent) {
    const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= n_nodes) return;
    const uint32_t count = recs[j].info >> 16, start = recs[j].tri_start;
    if (count == 0u || (recs[j].info == 0u)) return;
    for (uint32_t k = 0; k < count && start + k < n_ref; k++) { slot_rank[start + k] = recs[j].rank; slot_parent[start + k] = recs[j].parent; }
}
// Lists the references of triangles that lie in a face plane of the node holding them (see RtDeviceScene::risky_tris): list[0] counts,
// list[1 + k] are slots (as many as fit `cap`).  Runs while the leaves' box fields still hold their boxes.
// out[k] = tris[list[k]]
__global__ __launch_bounds__(BLOCK) void gather_records_kernel(const RtTriRec *__restrict__ tris, const uint32_t *__restrict__ list, uint32_t n, RtTriRec *__restrict__ out) {
    const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
    if (k < n) out[k] = tris[list[k]];
}

__global__ __launch_bounds__(BLOCK) void risky_kernel(const RtNodeRec *__restrict__ recs, uint32_t n_nodes, uint32_t n_ref, const uint4 *__restrict__ refs, const float4 *__restrict__ vertices,
                                                      uint32_t *__restrict__ list, uint32_t cap) {
    const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= n_nodes) return;
    const uint32_t count = recs[j].info >> 16, start = recs[j].tri_start;
    if (recs[j].info == 0u || count == 0u) return;
    for (uint32_t k = 0; k < count && start + k < n_ref; k++) {
        const uint4 t = refs[start + k];
        const float4 a = vertices[t.x], b = vertices[t.y], c = vertices[t.z];
        const float va[3] = {a.x, a.y, a.z}, vb[3] = {b.x, b.y, b.z}, vc[3] = {c.x, c.y, c.z};
        bool risky = false;
        for (int ax = 0; ax < 3; ax++)
            if (va[ax] == vb[ax] && vb[ax] == vc[ax] && (va[ax] == recs[j].lo[ax] || va[ax] == recs[j].hi[ax])) risky = true;
        if (risky) {
            const uint32_t at = atomicAdd(&list[0], 1u);
            if (at < cap) list[1u + at] = start + k;
        }
    }
}
__global__ __launch_bounds__(BLOCK) void iota_kernel(uint32_t *__restrict__ v, uint32_t n) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) v[i] = i;
}
// Sort keys of the references in their current order `perm`: which = 0: i1 << 32 | i2, which = 1: material << 32 | i0.
__global__ __launch_bounds__(BLOCK) void ref_key_kernel(const uint4 *__restrict__ refs, const uint32_t *__restrict__ perm, uint32_t n, int which, unsigned long long *__restrict__ key) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint4 t = refs[perm[i]];
    key[i] = which == 0 ? ((unsigned long long)t.y << 32) | t.z : ((unsigned long long)t.w << 32) | t.x;
}
// Position of the first reference of each run of equal triangles, as a running maximum over (position if a run starts there).
__global__ __launch_bounds__(BLOCK) void run_start_kernel(const uint4 *__restrict__ refs, const uint32_t *__restrict__ perm, uint32_t n, uint32_t *__restrict__ start) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    bool first = i == 0;
    if (!first) {
        const uint4 a = refs[perm[i]], b = refs[perm[i - 1]];
        first = a.x != b.x || a.y != b.y || a.z != b.z || a.w != b.w;
    }
    start[i] = first ? i : 0u;
}
__global__ __launch_bounds__(BLOCK) void dup_write_kernel(const uint32_t *__restrict__ perm, const uint32_t *__restrict__ start, const uint32_t *__restrict__ slot_parent, uint32_t n,
                                                          RtTriRec *__restrict__ tris, uint32_t *__restrict__ dup_parent) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t slot = perm[i];
    tris[slot].dup_first = start[i];
    tris[slot].dup_pos = i;
    dup_parent[i] = slot_parent[slot];
}

__global__ __launch_bounds__(BLOCK) void triangle_records_kernel(const float4 *__restrict__ vertices, const float4 *__restrict__ normals, const uint4 *__restrict__ refs, uint32_t n_ref,
                                                                 RtTriRec *__restrict__ tris, RtShadeRec *__restrict__ shade) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_ref) return;
    const uint4 t = refs[i];
    const float4 a = vertices[t.x], b = vertices[t.y], c = vertices[t.z];   // frag:469-471
    RtTriRec r;
    r.v1[0] = a.x; r.v1[1] = a.y; r.v1[2] = a.z;
    r.e1x = b.x - a.x; r.e1y = b.y - a.y; r.e1z = b.z - a.z;                // frag:256-257
    r.e2x = c.x - a.x; r.e2y = c.y - a.y; r.e2z = c.z - a.z;
    r.dup_first = r.dup_pos = i; r.pad = 0;
    tris[i] = r;
    const float4 n1 = normals[t.x], n2 = normals[t.y], n3 = normals[t.z];   // frag:543-545
    RtShadeRec h;
    h.a[0] = a.x; h.a[1] = a.y; h.a[2] = a.z;
    h.e1[0] = r.e1x; h.e1[1] = r.e1y; h.e1[2] = r.e1z;                      // frag:216-217: b - a, c - a
    h.e2[0] = r.e2x; h.e2[1] = r.e2y; h.e2[2] = r.e2z;
    h.n1[0] = n1.x; h.n1[1] = n1.y; h.n1[2] = n1.z;
    h.n2[0] = n2.x; h.n2[1] = n2.y; h.n2[2] = n2.z;
    h.n3[0] = n3.x; h.n3[1] = n3.y; h.n3[2] = n3.z;
    h.material = t.w; h.pad = 0;
    shade[i] = h;
}

__global__ __launch_bounds__(BLOCK) void leaf_triangle_kernel(RtNodeRec *__restrict__ nodes, uint32_t n_nodes, const RtTriRec *__restrict__ tris) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_nodes) return;
    const uint32_t info = nodes[i].info;
    if ((info & 0xFu) != 0u || (info >> 16) == 0u) return;
    const RtTriRec r = tris[nodes[i].tri_start];
    nodes[i].lo[0] = r.v1[0]; nodes[i].lo[1] = r.v1[1]; nodes[i].lo[2] = r.v1[2];
    nodes[i].mid[0] = r.e1x; nodes[i].mid[1] = r.e1y; nodes[i].mid[2] = r.e1z;
    nodes[i].hi[0] = r.e2x; nodes[i].hi[1] = r.e2y; nodes[i].hi[2] = r.e2z;
    nodes[i].first_id = r.dup_first;
}

// One plane record per distinct triangle (the first reference of its run) with a non-zero normal; `out` null: only counts.
__global__ __launch_bounds__(BLOCK) void plane_records_kernel(const RtTriRec *__restrict__ tris, uint32_t n_ref, uint32_t *__restrict__ count, RtPlaneRec *__restrict__ out) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_ref) return;
    const RtTriRec r = tris[i];
    if (r.dup_pos != r.dup_first) return;
    const float nx = r.e1y * r.e2z - r.e1z * r.e2y, ny = r.e1z * r.e2x - r.e1x * r.e2z, nz = r.e1x * r.e2y - r.e1y * r.e2x;
    const float len = sqrtf(nx * nx + ny * ny + nz * nz);
    if (!(len > 0.0f) || !(len < __builtin_inff())) return;
    const uint32_t k = atomicAdd(count, 1u);
    if (!out) return;
    RtPlaneRec p;
    p.n[0] = nx / len; p.n[1] = ny / len; p.n[2] = nz / len;
    p.v1x = r.v1[0]; p.v1y = r.v1[1]; p.v1z = r.v1[2];
    p.slot = i;
    p.edges = sqrtf(r.e1x * r.e1x + r.e1y * r.e1y + r.e1z * r.e1z) + sqrtf(r.e2x * r.e2x + r.e2y * r.e2y + r.e2z * r.e2z);
    out[k] = p;
}

inline uint32_t grid_for(size_t n) { return (uint32_t)((n + BLOCK - 1) / BLOCK); }

struct Buf {
    void *p = nullptr;
    Buf() = default;
    Buf(const Buf &) = delete;
    Buf &operator=(const Buf &) = delete;
    Buf(Buf &&o) noexcept : p(o.p) { o.p = nullptr; }
    ~Buf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { if (p) { (void)hipFree(p); p = nullptr; } return hipMalloc(&p, bytes ? bytes : 1); }
};

}  // namespace

#define REC_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { out->hip_error = (int)e_; return RT_ENODEVICE; } } while (0)

int rt_build_node_records(const uint32_t *d_words, size_t n_words, size_t n_ref, const float lo0[3], const float hi0[3], RtNodeBuild *out) {
    out->nodes = nullptr; out->n_nodes = 0; out->max_depth = 0; out->error = 0; out->error_word = 0; out->hip_error = 0;
    if (n_words == 0) return RT_OK;
    if (n_words < 2) { out->error = E_PAST_END; out->error_word = 0; return RT_ESCENE; }
    Buf level, next, tally, scan_tmp, error;
    std::vector<Buf> chunks;             // the records of every level, concatenated at the end
    std::vector<uint32_t> chunk_size;
    REC_TRY(error.alloc(8));
    REC_TRY(hipMemset(error.p, 0, 8));
    // level 0 is a block of eight like every other: the root and seven empty slots, so that all blocks start at multiples of 8
    Entry root[8];
    for (Entry &e : root) { e.word = HOLE; for (int k = 0; k < 3; k++) { e.lo[k] = 0.0f; e.hi[k] = 0.0f; } }
    root[0].word = 0u;
    for (int k = 0; k < 3; k++) { root[0].lo[k] = lo0[k]; root[0].hi[k] = hi0[k]; }
    REC_TRY(level.alloc(sizeof root));
    REC_TRY(hipMemcpy(level.p, root, sizeof root, hipMemcpyHostToDevice));
    uint32_t n = 8, base = 0, depth = 0;
    unsigned long long real_nodes = 1;   // a tree cannot hold more nodes than words: more means shared or cyclic children
    for (;; depth++) {
        REC_TRY(tally.alloc(((size_t)n + 1) * sizeof(Tally)));
        hipLaunchKernelGGL(check_nodes_kernel, dim3(grid_for((size_t)n + 1)), dim3(BLOCK), 0, nullptr, d_words, (uint32_t)n_words, (uint32_t)n_ref, (const Entry *)level.p, n, (Tally *)tally.p, (uint32_t *)error.p);
        REC_TRY(hipGetLastError());
        {
            size_t bytes = 0;
            REC_TRY(hipcub::DeviceScan::ExclusiveScan(nullptr, bytes, (const Tally *)tally.p, (Tally *)tally.p, AddTally(), Tally{0u, 0u}, (int)(n + 1), nullptr));
            REC_TRY(scan_tmp.alloc(bytes));
            REC_TRY(hipcub::DeviceScan::ExclusiveScan(scan_tmp.p, bytes, (const Tally *)tally.p, (Tally *)tally.p, AddTally(), Tally{0u, 0u}, (int)(n + 1), nullptr));
        }
        Tally total{0u, 0u};
        uint32_t err[2] = {0u, 0u};
        REC_TRY(hipMemcpy(&total, (const Tally *)tally.p + n, sizeof total, hipMemcpyDeviceToHost));
        REC_TRY(hipMemcpy(err, error.p, sizeof err, hipMemcpyDeviceToHost));
        if (err[0]) { out->error = err[0]; out->error_word = err[1]; return RT_ESCENE; }
        real_nodes += total.children;
        if (total.blocks && depth + 1 > RT_MAX_TREE_DEPTH) { out->error = RT_NODE_ERROR_TOO_DEEP; return RT_ESCENE; }
        if (real_nodes > n_words) { out->error = RT_NODE_ERROR_TOO_MANY; return RT_ESCENE; }
        const unsigned long long next_n = 8ull * total.blocks;
        if ((unsigned long long)base + n + next_n > RT_MAX_NODE_RECORDS) { out->error = RT_NODE_ERROR_RECORD_LIMIT; return RT_ELIMIT; }
        chunks.emplace_back();
        chunk_size.push_back(n);
        REC_TRY(chunks.back().alloc((size_t)n * sizeof(RtNodeRec)));
        REC_TRY(next.alloc((size_t)next_n * sizeof(Entry)));
        hipLaunchKernelGGL(write_nodes_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, nullptr, d_words, (const Entry *)level.p, n, (const Tally *)tally.p, base + n, (RtNodeRec *)chunks.back().p, (Entry *)next.p);
        REC_TRY(hipGetLastError());
        out->max_depth = depth;
        base += n;
        if (next_n == 0) break;
        REC_TRY(hipDeviceSynchronize());      // `level` is released below while the kernel may still read it
        std::swap(level.p, next.p);
        n = (uint32_t)next_n;
    }
    void *all = nullptr;
    REC_TRY(hipMalloc(&all, (size_t)base * sizeof(RtNodeRec)));
    size_t at = 0;
    for (size_t l = 0; l < chunks.size(); l++) {
        hipError_t e = hipMemcpy((char *)all + at, chunks[l].p, (size_t)chunk_size[l] * sizeof(RtNodeRec), hipMemcpyDeviceToDevice);
        if (e != hipSuccess) { (void)hipFree(all); out->hip_error = (int)e; return RT_ENODEVICE; }
        at += (size_t)chunk_size[l] * sizeof(RtNodeRec);
    }
    // Rank of every node in the reference traversal's order: subtree sizes deepest level first, then numbering from the root.
    {
        Buf size;
        hipError_t e = size.alloc((size_t)base * sizeof(uint32_t));
        std::vector<uint32_t> first(chunk_size.size());
        uint32_t off = 0;
        for (size_t l = 0; l < chunk_size.size(); l++) { first[l] = off; off += chunk_size[l]; }
        RtNodeRec *recs = (RtNodeRec *)all;
        for (size_t l = chunk_size.size(); e == hipSuccess && l-- > 0;) {
            hipLaunchKernelGGL(subtree_size_kernel, dim3(grid_for(chunk_size[l])), dim3(BLOCK), 0, nullptr, (const RtNodeRec *)recs, first[l], chunk_size[l], (uint32_t *)size.p, l == 0);
            e = hipGetLastError();
        }
        const uint32_t no_parent = RT_NO_PARENT;
        if (e == hipSuccess) e = hipMemcpy(&recs[0].parent, &no_parent, sizeof no_parent, hipMemcpyHostToDevice);   // (root rank: 0 from the record's initialisation)
        for (size_t l = 0; e == hipSuccess && l + 1 < chunk_size.size(); l++) {
            hipLaunchKernelGGL(child_rank_kernel, dim3(grid_for(chunk_size[l])), dim3(BLOCK), 0, nullptr, recs, first[l], chunk_size[l], (const uint32_t *)size.p);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) { (void)hipFree(all); out->hip_error = (int)e; return RT_ENODEVICE; }
    }
    out->nodes = (RtNodeRec *)all;
    out->n_nodes = base;
    return RT_OK;
}

int rt_build_scene_records(const float *vertices, const float *normals, const uint32_t *tri_refs, uint32_t n_ref, RtNodeRec *nodes, uint32_t n_nodes, RtTriRec *tris, RtShadeRec *shade,
                           uint32_t *dup_parent, RtTriRec **risky, uint32_t *n_risky) {
    *risky = nullptr; *n_risky = 0;
    if (n_ref) hipLaunchKernelGGL(triangle_records_kernel, dim3(grid_for(n_ref)), dim3(BLOCK), 0, nullptr, (const float4 *)vertices, (const float4 *)normals, (const uint4 *)tri_refs, n_ref, tris, shade);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || n_ref == 0) return (int)e;
    // The references of each triangle, sorted by the rank of the node that holds them (see RtTriRec): a stable radix sort by
    // rank, then by (i1, i2), then by (material, i0) leaves equal triangles adjacent and in rank order.
    Buf slot_rank, slot_parent, perm_a, perm_b, key_a, key_b, tmp, start;
    const size_t n = n_ref;
    if ((e = slot_rank.alloc(n * 4)) != hipSuccess || (e = slot_parent.alloc(n * 4)) != hipSuccess || (e = perm_a.alloc(n * 4)) != hipSuccess || (e = perm_b.alloc(n * 4)) != hipSuccess ||
        (e = key_a.alloc(n * 8)) != hipSuccess || (e = key_b.alloc(n * 8)) != hipSuccess || (e = start.alloc(n * 4)) != hipSuccess) return (int)e;
    if ((e = hipMemset(slot_rank.p, 0xFF, n * 4)) != hipSuccess || (e = hipMemset(slot_parent.p, 0xFF, n * 4)) != hipSuccess) return (int)e;
    if (n_nodes) hipLaunchKernelGGL(slot_owner_kernel, dim3(grid_for(n_nodes)), dim3(BLOCK), 0, nullptr, (const RtNodeRec *)nodes, n_nodes, n_ref, (uint32_t *)slot_rank.p, (uint32_t *)slot_parent.p);
    hipLaunchKernelGGL(iota_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, nullptr, (uint32_t *)perm_a.p, n_ref);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    size_t bytes32 = 0, bytes64 = 0;
    if ((e = hipcub::DeviceRadixSort::SortPairs(nullptr, bytes32, (const uint32_t *)slot_rank.p, (uint32_t *)key_a.p, (const uint32_t *)perm_a.p, (uint32_t *)perm_b.p, (int)n)) != hipSuccess) return (int)e;
    if ((e = hipcub::DeviceRadixSort::SortPairs(nullptr, bytes64, (const unsigned long long *)key_a.p, (unsigned long long *)key_b.p, (const uint32_t *)perm_a.p, (uint32_t *)perm_b.p, (int)n)) != hipSuccess) return (int)e;
    if ((e = tmp.alloc(bytes32 > bytes64 ? bytes32 : bytes64)) != hipSuccess) return (int)e;
    if ((e = hipcub::DeviceRadixSort::SortPairs(tmp.p, bytes32, (const uint32_t *)slot_rank.p, (uint32_t *)key_a.p, (const uint32_t *)perm_a.p, (uint32_t *)perm_b.p, (int)n)) != hipSuccess) return (int)e;
    uint32_t *cur = (uint32_t *)perm_b.p, *other = (uint32_t *)perm_a.p;
    for (int which = 0; which < 2; which++) {
        hipLaunchKernelGGL(ref_key_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, nullptr, (const uint4 *)tri_refs, (const uint32_t *)cur, n_ref, which, (unsigned long long *)key_a.p);
        if ((e = hipGetLastError()) != hipSuccess) return (int)e;
        if ((e = hipcub::DeviceRadixSort::SortPairs(tmp.p, bytes64, (const unsigned long long *)key_a.p, (unsigned long long *)key_b.p, (const uint32_t *)cur, other, (int)n)) != hipSuccess) return (int)e;
        std::swap(cur, other);
    }
    hipLaunchKernelGGL(run_start_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, nullptr, (const uint4 *)tri_refs, (const uint32_t *)cur, n_ref, (uint32_t *)start.p);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    {
        size_t sb = 0;
        if ((e = hipcub::DeviceScan::InclusiveScan(nullptr, sb, (const uint32_t *)start.p, (uint32_t *)start.p, hipcub::Max(), (int)n)) != hipSuccess) return (int)e;
        Buf scan_tmp;
        if ((e = scan_tmp.alloc(sb)) != hipSuccess) return (int)e;
        if ((e = hipcub::DeviceScan::InclusiveScan(scan_tmp.p, sb, (const uint32_t *)start.p, (uint32_t *)start.p, hipcub::Max(), (int)n)) != hipSuccess) return (int)e;
        if ((e = hipDeviceSynchronize()) != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(dup_write_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, nullptr, (const uint32_t *)cur, (const uint32_t *)start.p, (const uint32_t *)slot_parent.p, n_ref, tris, dup_parent);
    // triangles in a face plane of their node: the distinct ones (by RtTriRec::dup_first), if they are few
    if (n_nodes) {
        constexpr uint32_t CAP = 4096;
        Buf list;
        if ((e = list.alloc((CAP + 1) * 4)) != hipSuccess || (e = hipMemset(list.p, 0, 4)) != hipSuccess) return (int)e;
        hipLaunchKernelGGL(risky_kernel, dim3(grid_for(n_nodes)), dim3(BLOCK), 0, nullptr, (const RtNodeRec *)nodes, n_nodes, n_ref, (const uint4 *)tri_refs, (const float4 *)vertices, (uint32_t *)list.p, CAP);
        std::vector<uint32_t> h(CAP + 1);
        if ((e = hipMemcpy(h.data(), list.p, (CAP + 1) * 4, hipMemcpyDeviceToHost)) != hipSuccess) return (int)e;
        if (h[0] > CAP) *n_risky = RT_RISKY_TOO_MANY;
        else if (h[0]) {
            // the listed records are gathered into one buffer on the device and copied once (a copy per record costs tens of
            // microseconds each, and a triangle is listed once per node that holds it in a face plane)
            Buf gathered;
            if ((e = gathered.alloc((size_t)h[0] * sizeof(RtTriRec))) != hipSuccess) return (int)e;
            hipLaunchKernelGGL(gather_records_kernel, dim3(grid_for(h[0])), dim3(BLOCK), 0, nullptr, (const RtTriRec *)tris, (const uint32_t *)list.p + 1, h[0], (RtTriRec *)gathered.p);
            if ((e = hipGetLastError()) != hipSuccess) return (int)e;
            std::vector<RtTriRec> all(h[0]), recs;
            if ((e = hipMemcpy(all.data(), gathered.p, (size_t)h[0] * sizeof(RtTriRec), hipMemcpyDeviceToHost)) != hipSuccess) return (int)e;
            for (uint32_t k = 0; k < h[0] && *n_risky != RT_RISKY_TOO_MANY; k++) {
                const RtTriRec &r = all[k];
                bool seen = false;
                for (const RtTriRec &q : recs) seen = seen || q.dup_first == r.dup_first;
                if (!seen) { if (recs.size() == RT_MAX_RISKY) *n_risky = RT_RISKY_TOO_MANY; else recs.push_back(r); }
            }
            if (*n_risky != RT_RISKY_TOO_MANY) {
                if ((e = hipMalloc((void **)risky, recs.size() * sizeof(RtTriRec))) != hipSuccess) return (int)e;
                if ((e = hipMemcpy(*risky, recs.data(), recs.size() * sizeof(RtTriRec), hipMemcpyHostToDevice)) != hipSuccess) return (int)e;
                *n_risky = (uint32_t)recs.size();
            }
        }
    }
    if (n_nodes) hipLaunchKernelGGL(leaf_triangle_kernel, dim3(grid_for(n_nodes)), dim3(BLOCK), 0, nullptr, nodes, n_nodes, (const RtTriRec *)tris);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    return (int)e;
}

int rt_build_plane_records(const RtTriRec *tris, uint32_t n_ref, RtPlaneRec **out, uint32_t *n_out) {
    *out = nullptr; *n_out = 0;
    if (n_ref == 0) return 0;
    Buf count;
    hipError_t e;
    uint32_t n = 0;
    if ((e = count.alloc(4)) != hipSuccess || (e = hipMemset(count.p, 0, 4)) != hipSuccess) return (int)e;
    hipLaunchKernelGGL(plane_records_kernel, dim3(grid_for(n_ref)), dim3(BLOCK), 0, nullptr, tris, n_ref, (uint32_t *)count.p, (RtPlaneRec *)nullptr);
    if ((e = hipGetLastError()) != hipSuccess || (e = hipMemcpy(&n, count.p, 4, hipMemcpyDeviceToHost)) != hipSuccess) return (int)e;
    if (n == 0) return 0;
    if ((e = hipMalloc((void **)out, (size_t)n * sizeof(RtPlaneRec))) != hipSuccess) return (int)e;
    if ((e = hipMemset(count.p, 0, 4)) != hipSuccess) return (int)e;
    hipLaunchKernelGGL(plane_records_kernel, dim3(grid_for(n_ref)), dim3(BLOCK), 0, nullptr, tris, n_ref, (uint32_t *)count.p, *out);
    if ((e = hipGetLastError()) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess) { (void)hipFree(*out); *out = nullptr; return (int)e; }
    *n_out = n;
    return 0;
}
