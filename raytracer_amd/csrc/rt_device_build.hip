// rt_octree_build_device (include/rt_hotpath.h): the reference's octree construction (app/core/src/Octree.cpp:118-256)
// with the work that grows with the scene -- the triangle/box overlap tests of Subdivide (Octree.cpp:146-170,
// app/core/include/TriangleBoxIntersection.hpp:153-240) and the distribution of the triangle references to the
// children -- on the GPU.  The result is bit-identical to rt_octree_build's.
//
// The tree is built breadth-first, one level per pass.  A level is a list of (node, triangle) pairs, grouped by node in
// node order and, within a node, in the order the reference's std::vector holds them:
//   classify_kernel   one lane per pair: the eight child overlap tests -> overlap mask and count
//   (prefix sums)     an 8-wide exclusive sum over the masks of the pairs that go down (rocPRIM): the rank of a pair
//                     among the earlier pairs of its node that enter the same child = its place in that child's list
//   child_sizes_kernel per node and octant the size of the child's list; a second prefix sum over these gives every
//                     non-empty child its index and its first pair in the next level
//   next_level_kernel / scatter_kernel / tail_kernel   write the next level's segments, boxes and pairs
// Triangles that overlap `keep_after` or more children are the reference's "keep" candidates (Octree.cpp:167-181): at
// most max_per_node of them stay in the node, chosen by std::partial_sort on the overlap count alone, whose order among
// equal counts is whatever the library's heap leaves.  They are few, so they are copied to the host, ordered there by
// the same library call on the same sequence, and the ones pushed down are appended to the children's lists
// (tail_kernel) exactly where the reference appends them.  Per-node data never visits the host otherwise: the
// flattening (Octree.cpp:38-85) is a bottom-up pass for the sizes of the subtrees (subtree_kernel) and a top-down pass
// that places every node's words and references in pre-order (place_kernel, emit_kernel, emit_kept_kernel); the two
// flattened arrays come back in one copy each.
//
// Built WITHOUT -fgpu-flush-denormals-to-zero: the reference's builder runs on the CPU with denormals preserved.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "rt_build_internal.h"

// rt_api.cpp: scene creation from arrays already in device memory; this thread's rt_set_device choice (-1: none) swapped for another
extern "C" int rt_scene_create_from(const float *vertices, const float *normals, size_t n_vert, const uint32_t *tri_refs, size_t n_ref,
                                    const uint32_t *nodes, size_t n_words, const float bounds_min[3], const float bounds_max[3], int source_is_device, rt_scene **out);
extern "C" int rt_internal_swap_device(int device);

namespace {

using rt_host::Tri;
constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr uint32_t CAND_BIT = 0x8000u;     // class word: overlap mask | count << 8 | CAND_BIT
constexpr int BLOCK = 256;

struct Box { float lo[3], hi[3]; };
struct Count8 { uint32_t c[8]; };
struct CandRec { uint32_t pair, node, cls; Tri tri; };     // a keep candidate, copied to the host
struct PushRec { uint32_t dest, node; Tri tri; };          // (rank, slot or node, triangle) of a candidate pushed down or kept

struct AddCount8 {
    __host__ __device__ Count8 operator()(const Count8 &a, const Count8 &b) const {
        Count8 r;
        for (int i = 0; i < 8; i++) r.c[i] = a.c[i] + b.c[i];
        return r;
    }
};
struct GoesDown {      // class word -> one count per octant, for the pairs that are handed to the children in list order
    __host__ __device__ Count8 operator()(uint16_t cls) const {
        Count8 r;
        const uint32_t m = (cls & CAND_BIT) ? 0u : (cls & 0xFFu);
        for (int i = 0; i < 8; i++) r.c[i] = (m >> i) & 1u;
        return r;
    }
};
struct Place { uint32_t first_pair, index; uint32_t overflow = 0; };   // of a child slot: first pair and node index in the next level
struct AddPlace {
    __host__ __device__ Place operator()(const Place &a, const Place &b) const {
        Place r;
        const unsigned long long p = (unsigned long long)a.first_pair + b.first_pair;
        r.first_pair = (uint32_t)p; r.index = a.index + b.index;
        r.overflow = a.overflow | b.overflow | (p >= 0x7FFFFFFFull ? 1u : 0u);
        return r;
    }
};
struct SizeToPlace { __host__ __device__ Place operator()(uint32_t size) const { Place r; r.first_pair = size; r.index = size ? 1u : 0u; r.overflow = 0; return r; } };
struct IsCandidate { __host__ __device__ uint8_t operator()(uint16_t cls) const { return (uint8_t)((cls & CAND_BIT) ? 1 : 0); } };

// Octree.cpp:119-124: a node is split unless it sits at the depth limit or holds few enough triangles
__device__ inline bool node_splits(uint32_t size, uint32_t max_per_leaf, uint32_t depth_allows_split) { return depth_allows_split && size > max_per_leaf; }

__global__ __launch_bounds__(BLOCK) void classify_kernel(const Tri *__restrict__ pairs, const uint32_t *__restrict__ node_of, const Box *__restrict__ boxes,
                                                         const uint32_t *__restrict__ seg, const float4 *__restrict__ verts, uint32_t n_pairs,
                                                         uint32_t keep_after, uint32_t max_per_leaf, uint32_t depth_allows_split, uint16_t *__restrict__ cls) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_pairs) return;
    const uint32_t node = node_of[i];
    if (!node_splits(seg[node + 1] - seg[node], max_per_leaf, depth_allows_split)) { cls[i] = 0; return; }
    const Tri t = pairs[i];
    const float4 a4 = verts[t.a], b4 = verts[t.b], c4 = verts[t.c];
    const float a[3] = {a4.x, a4.y, a4.z}, b[3] = {b4.x, b4.y, b4.z}, c[3] = {c4.x, c4.y, c4.z};
    const Box bx = boxes[node];
    uint32_t mask = 0, count = 0;
    for (int o = 0; o < 8; o++) {
        float clo[3], chi[3], centre[3], half[3];
        rt_host::child_box(bx.lo, bx.hi, o, clo, chi);
        rt_host::box_centre_half(clo, chi, centre, half);
        if (rt_host::triangle_overlaps_box(centre, half, a, b, c)) { mask |= 1u << o; count++; }
    }
    cls[i] = (uint16_t)(mask | (count << 8) | (count >= keep_after ? CAND_BIT : 0u));   // Octree.cpp:163
}

// size of every child's list before the pushed-down candidates are added: the pairs of the node that enter octant o
__global__ __launch_bounds__(BLOCK) void child_sizes_kernel(const Count8 *__restrict__ rank, const uint32_t *__restrict__ seg, uint32_t n_nodes, uint32_t *__restrict__ down,
                                                            uint32_t *__restrict__ sizes) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i > n_nodes * 8u) return;
    if (i == n_nodes * 8u) { sizes[i] = 0; return; }      // the prefix sum runs over one entry more
    const uint32_t node = i >> 3, o = i & 7u;
    const uint32_t n = rank[seg[node + 1]].c[o] - rank[seg[node]].c[o];
    down[i] = n;
    sizes[i] = n;
}

// pushed-down candidates: (node * 8 + octant, how many), every slot at most once
__global__ __launch_bounds__(BLOCK) void add_pushed_kernel(const uint2 *__restrict__ add, uint32_t n, uint32_t *__restrict__ sizes) {
    const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
    if (j < n) sizes[add[j].x] += add[j].y;
}

__global__ __launch_bounds__(BLOCK) void candidates_kernel(const uint32_t *__restrict__ cand_idx, uint32_t n_cand, const Tri *__restrict__ pairs,
                                                           const uint32_t *__restrict__ node_of, const uint16_t *__restrict__ cls, CandRec *__restrict__ out) {
    const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= n_cand) return;
    const uint32_t i = cand_idx[j];
    out[j] = CandRec{i, node_of[i], cls[i], pairs[i]};
}

__global__ __launch_bounds__(BLOCK) void scatter_kernel(const Tri *__restrict__ pairs, const uint32_t *__restrict__ node_of, const uint16_t *__restrict__ cls,
                                                        const Count8 *__restrict__ rank, const uint32_t *__restrict__ seg, const Place *__restrict__ place,
                                                        uint32_t n_pairs, Tri *__restrict__ next_pairs, uint32_t *__restrict__ next_node_of) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_pairs) return;
    const uint32_t c = cls[i];
    if ((c & CAND_BIT) || (c & 0xFFu) == 0u) return;
    const uint32_t node = node_of[i];
    const Tri t = pairs[i];
    const Count8 mine = rank[i], first = rank[seg[node]];
    for (uint32_t o = 0; o < 8; o++)
        if (c & (1u << o)) {
            const Place pl = place[node * 8u + o];
            const uint32_t dest = pl.first_pair + (mine.c[o] - first.c[o]);
            next_pairs[dest] = t;
            next_node_of[dest] = pl.index;
        }
}

__global__ __launch_bounds__(BLOCK) void tail_kernel(const PushRec *__restrict__ recs, uint32_t n, const uint32_t *__restrict__ down, const Place *__restrict__ place,
                                                     Tri *__restrict__ next_pairs, uint32_t *__restrict__ next_node_of) {
    const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= n) return;
    const PushRec r = recs[j];            // r.dest: rank among the pushed-down candidates of the slot, r.node: the slot
    const uint32_t dest = place[r.node].first_pair + down[r.node] + r.dest;
    next_pairs[dest] = r.tri;
    next_node_of[dest] = place[r.node].index;
}

// Every non-empty child becomes a node of the next level (Octree.cpp:194-198: empty children are never visited and are
// dropped when flattening): its segment start, its box (Octree.cpp:130-142) and the parent's link to it.
__global__ __launch_bounds__(BLOCK) void next_level_kernel(const Box *__restrict__ boxes, const uint32_t *__restrict__ sizes, const Place *__restrict__ place, uint32_t n_nodes,
                                                           uint32_t *__restrict__ kid, uint32_t *__restrict__ next_seg, Box *__restrict__ next_boxes) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i > n_nodes * 8u) return;
    if (i == n_nodes * 8u) { next_seg[place[i].index] = place[i].first_pair; return; }     // totals: closes the last segment
    if (sizes[i] == 0) { kid[i] = NONE; return; }
    const uint32_t child = place[i].index;
    kid[i] = child;
    next_seg[child] = place[i].first_pair;
    const Box b = boxes[i >> 3];
    Box out;
    rt_host::child_box(b.lo, b.hi, (int)(i & 7u), out.lo, out.hi);
    next_boxes[child] = out;
}

// ---- flattening (Octree.cpp:38-85) -------------------------------------------------------------------------------
// own[node]: triangles the node itself holds in the flattened tree: what it kept if it was split, its whole list if not
__global__ __launch_bounds__(BLOCK) void own_kernel(const uint32_t *__restrict__ seg, uint32_t n_nodes, uint32_t max_per_leaf, uint32_t depth_allows_split, uint32_t *__restrict__ own) {
    const uint32_t n = blockIdx.x * BLOCK + threadIdx.x;
    if (n >= n_nodes) return;
    const uint32_t size = seg[n + 1] - seg[n];
    own[n] = node_splits(size, max_per_leaf, depth_allows_split) ? 0u : size;
}
__global__ __launch_bounds__(BLOCK) void kept_counts_kernel(const uint2 *__restrict__ kept, uint32_t n, uint32_t *__restrict__ own) {
    const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
    if (j < n) own[kept[j].x] = kept[j].y;
}
// words and references of the subtree below (and including) every node; kid == nullptr: the deepest level
__global__ __launch_bounds__(BLOCK) void subtree_kernel(const uint32_t *__restrict__ own, const uint32_t *__restrict__ kid, const uint2 *__restrict__ below, uint32_t n_nodes,
                                                        uint2 *__restrict__ sub, uint32_t *__restrict__ overflow) {
    const uint32_t n = blockIdx.x * BLOCK + threadIdx.x;
    if (n >= n_nodes) return;
    unsigned long long words = 2, refs = own[n];
    if (kid)
        for (uint32_t o = 0; o < 8; o++) {
            const uint32_t k = kid[n * 8u + o];
            if (k != NONE) { words += 1ull + below[k].x; refs += below[k].y; }
        }
    if (words > 0xFFFFFFFFull || refs > 0xFFFFFFFFull) *overflow = 1u;
    sub[n] = make_uint2((uint32_t)words, (uint32_t)refs);
}
// Pre-order: a node's header and child pointers, then its children's subtrees in ascending octant order; its references,
// then its children's.  at[node] = (first word, first reference) of the node; writes the children's.
__global__ __launch_bounds__(BLOCK) void place_kernel(const uint32_t *__restrict__ own, const uint32_t *__restrict__ kid, const uint2 *__restrict__ below, const uint2 *__restrict__ at,
                                                      uint32_t n_nodes, uint2 *__restrict__ at_below, uint32_t *__restrict__ words) {
    const uint32_t n = blockIdx.x * BLOCK + threadIdx.x;
    if (n >= n_nodes) return;
    const uint2 me = at[n];
    uint32_t mask = 0, count = 0;
    if (kid) for (uint32_t o = 0; o < 8; o++) if (kid[n * 8u + o] != NONE) { mask |= 1u << o; count++; }
    words[me.x] = ((own[n] & 0xFFFFu) << 16) | (mask << 8) | count;     // 16-bit triangle count (Octree.cpp:44)
    words[me.x + 1] = me.y;
    uint32_t w = me.x + 2u + count, r = me.y + own[n], j = 0;
    if (kid)
        for (uint32_t o = 0; o < 8; o++) {
            const uint32_t k = kid[n * 8u + o];
            if (k == NONE) continue;
            words[me.x + 2u + j++] = w;
            at_below[k] = make_uint2(w, r);
            w += below[k].x; r += below[k].y;
        }
}
// references of the nodes that were not split: their pair list, in order
__global__ __launch_bounds__(BLOCK) void emit_kernel(const Tri *__restrict__ pairs, const uint32_t *__restrict__ node_of, const uint32_t *__restrict__ seg, const uint2 *__restrict__ at,
                                                     uint32_t n_pairs, uint32_t max_per_leaf, uint32_t depth_allows_split, Tri *__restrict__ refs) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_pairs) return;
    const uint32_t node = node_of[i], first = seg[node];
    if (!node_splits(seg[node + 1] - first, max_per_leaf, depth_allows_split)) refs[at[node].y + (i - first)] = pairs[i];
}
// references of the split nodes: the candidates they kept (r.node: the node, r.dest: rank in its kept list)
__global__ __launch_bounds__(BLOCK) void emit_kept_kernel(const PushRec *__restrict__ kept, uint32_t n, const uint2 *__restrict__ at, Tri *__restrict__ refs) {
    const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
    if (j < n) refs[at[kept[j].node].y + kept[j].dest] = kept[j].tri;
}

// ---- host side ------------------------------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) {         // grows only: the per-level scratch buffers are reused from level to level
        if (bytes <= cap && p) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        const size_t want = bytes ? bytes + bytes / 4 : 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    template <class T> T *as() const { return (T *)p; }
};

struct Level {                       // what stays on the device of one tree level until the tree is flattened
    uint32_t n_nodes = 0, n_pairs = 0;
    bool depth_allows_split = false;
    DevBuf pairs, node_of, seg, kid, own, sub, at;
    std::vector<PushRec> kept;       // the triangles split nodes kept: (node, rank, triangle)
};

inline uint32_t grid_for(size_t n) { return (uint32_t)((n + BLOCK - 1) / BLOCK); }

#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { char m_[256]; snprintf(m_, sizeof m_, "%s: %s", #call, hipGetErrorString(e_)); return rt_internal_set_error(RT_ENODEVICE, m_); } } while (0)

struct Cand { uint8_t mask, count; uint32_t slot; };   // slot: index into the level's candidate records

struct PhaseClock {          // RT_BUILD_TRACE=1: wall time per phase on stderr
    bool on = getenv("RT_BUILD_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    double acc[6] = {0, 0, 0, 0, 0, 0};
    void lap(int k) { if (!on) return; auto n = std::chrono::steady_clock::now(); acc[k] += std::chrono::duration<double, std::milli>(n - t).count(); t = n; }
    void report() { if (on) fprintf(stderr, "device octree build ms: upload %.1f | classify+rank+candidates %.1f | candidate ordering (host) %.1f | next level %.1f | flatten %.1f | copy back %.1f\n", acc[0], acc[1], acc[2], acc[3], acc[4], acc[5]); }
};

// scene != nullptr: instead of copying the flattened arrays to the host (`out` is then only given the sizes, bounds and depth), the scene is
// created from them where they are.
int build_on_device(const float *verts, const float *norms, size_t n_vert, const std::vector<Tri> &all, const rt_host::Params &P, rt_octree *out, rt_scene **scene = nullptr) {
    PhaseClock clock;
    float root_lo[3], root_hi[3];
    rt_host::root_bounds(verts, all, root_lo, root_hi);
    if (all.size() >= 0x7FFFFFFFull) return rt_internal_set_error(RT_ELIMIT, "too many triangles for the device builder");
    hipStream_t st = nullptr;
    HIP_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); } } sg{st};

    DevBuf d_verts;
    HIP_OK(d_verts.alloc(n_vert * 16));
    if (n_vert) HIP_OK(hipMemcpyAsync(d_verts.p, verts, n_vert * 16, hipMemcpyHostToDevice, st));

    std::vector<std::unique_ptr<Level>> levels;
    DevBuf boxes, next_boxes;
    {   // level 0: the root holds every triangle (Octree.cpp:248-250)
        auto L = std::make_unique<Level>();
        L->n_nodes = 1; L->n_pairs = (uint32_t)all.size();
        const uint32_t seg0[2] = {0u, L->n_pairs};
        Box rb;
        for (int k = 0; k < 3; k++) { rb.lo[k] = root_lo[k]; rb.hi[k] = root_hi[k]; }
        HIP_OK(L->pairs.alloc((size_t)L->n_pairs * sizeof(Tri)));
        HIP_OK(L->node_of.alloc((size_t)L->n_pairs * 4));
        HIP_OK(L->seg.alloc(8));
        HIP_OK(boxes.alloc(sizeof(Box)));
        if (L->n_pairs) {
            HIP_OK(hipMemcpyAsync(L->pairs.p, all.data(), (size_t)L->n_pairs * sizeof(Tri), hipMemcpyHostToDevice, st));
            HIP_OK(hipMemsetAsync(L->node_of.p, 0, (size_t)L->n_pairs * 4, st));
        }
        HIP_OK(hipMemcpyAsync(L->seg.p, seg0, 8, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(boxes.p, &rb, sizeof rb, hipMemcpyHostToDevice, st));
        HIP_OK(hipStreamSynchronize(st));      // seg0 and rb are stack objects
        levels.push_back(std::move(L));
    }
    clock.lap(0);

    size_t max_depth = 0;
    const uint32_t keep_after = (uint32_t)std::min<size_t>(P.keep_after, 9), max_leaf = (uint32_t)std::min<size_t>(P.max_per_leaf, 0xFFFFFFFFu);
    DevBuf cls, rank, down, sizes, place, cand_idx, cand_n, cand_recs, scan_tmp, add_d, push_d, kept_d;
    std::vector<CandRec> cand_host;
    std::vector<PushRec> pushes;
    std::vector<uint2> adds, kept_counts;
    std::vector<Cand> cands;
    HIP_OK(cand_n.alloc(4));
    for (uint32_t depth = 0;; depth++) {
        Level &L = *levels.back();
        const uint32_t n_nodes = L.n_nodes, n_pairs = L.n_pairs;
        // Octree.cpp:119-125: m_max_depth is raised for every visited node below the depth limit
        L.depth_allows_split = (size_t)depth + 1 < P.depth_limit;
        if (L.depth_allows_split) max_depth = std::max(max_depth, (size_t)depth + 1);
        HIP_OK(L.own.alloc((size_t)n_nodes * 4));
        own_kernel<<<grid_for(n_nodes), BLOCK, 0, st>>>(L.seg.as<uint32_t>(), n_nodes, max_leaf, L.depth_allows_split ? 1u : 0u, L.own.as<uint32_t>());
        HIP_OK(hipGetLastError());
        if (!L.depth_allows_split || n_pairs == 0) break;

        // ---- classify, rank, child sizes ---------------------------------------------------------------------------
        HIP_OK(cls.alloc(((size_t)n_pairs + 1) * 2));
        HIP_OK(hipMemsetAsync((char *)cls.p + (size_t)n_pairs * 2, 0, 2, st));     // the sum runs over n_pairs + 1 entries
        classify_kernel<<<grid_for(n_pairs), BLOCK, 0, st>>>(L.pairs.as<Tri>(), L.node_of.as<uint32_t>(), boxes.as<Box>(), L.seg.as<uint32_t>(), d_verts.as<float4>(),
                                                              n_pairs, keep_after, max_leaf, 1u, cls.as<uint16_t>());
        HIP_OK(hipGetLastError());
        HIP_OK(rank.alloc(((size_t)n_pairs + 1) * sizeof(Count8)));
        {
            hipcub::TransformInputIterator<Count8, GoesDown, const uint16_t *> in(cls.as<uint16_t>(), GoesDown());
            size_t bytes = 0;
            HIP_OK(hipcub::DeviceScan::ExclusiveScan(nullptr, bytes, in, rank.as<Count8>(), AddCount8(), Count8{}, (int)(n_pairs + 1), st));
            HIP_OK(scan_tmp.alloc(bytes));
            HIP_OK(hipcub::DeviceScan::ExclusiveScan(scan_tmp.p, bytes, in, rank.as<Count8>(), AddCount8(), Count8{}, (int)(n_pairs + 1), st));
        }
        const size_t slots = (size_t)n_nodes * 8;
        if (slots + 1 >= 0x7FFFFFFFull) return rt_internal_set_error(RT_ELIMIT, "a tree level has too many nodes for the device builder");
        HIP_OK(down.alloc(slots * 4));
        HIP_OK(sizes.alloc((slots + 1) * 4));
        child_sizes_kernel<<<grid_for(slots + 1), BLOCK, 0, st>>>(rank.as<Count8>(), L.seg.as<uint32_t>(), n_nodes, down.as<uint32_t>(), sizes.as<uint32_t>());
        HIP_OK(hipGetLastError());
        // ---- the keep candidates, in pair order, to the host ---------------------------------------------------------
        HIP_OK(cand_idx.alloc((size_t)n_pairs * 4));
        {
            hipcub::TransformInputIterator<uint8_t, IsCandidate, const uint16_t *> flags(cls.as<uint16_t>(), IsCandidate());
            hipcub::CountingInputIterator<uint32_t> idx(0u);
            size_t bytes = 0;
            HIP_OK(hipcub::DeviceSelect::Flagged(nullptr, bytes, idx, flags, cand_idx.as<uint32_t>(), cand_n.as<uint32_t>(), (int)n_pairs, st));
            HIP_OK(scan_tmp.alloc(bytes));
            HIP_OK(hipcub::DeviceSelect::Flagged(scan_tmp.p, bytes, idx, flags, cand_idx.as<uint32_t>(), cand_n.as<uint32_t>(), (int)n_pairs, st));
        }
        uint32_t n_cand = 0;
        HIP_OK(hipMemcpyAsync(&n_cand, cand_n.p, 4, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        cand_host.resize(n_cand);
        if (n_cand) {
            HIP_OK(cand_recs.alloc((size_t)n_cand * sizeof(CandRec)));
            candidates_kernel<<<grid_for(n_cand), BLOCK, 0, st>>>(cand_idx.as<uint32_t>(), n_cand, L.pairs.as<Tri>(), L.node_of.as<uint32_t>(), cls.as<uint16_t>(), cand_recs.as<CandRec>());
            HIP_OK(hipGetLastError());
            HIP_OK(hipMemcpyAsync(cand_host.data(), cand_recs.p, (size_t)n_cand * sizeof(CandRec), hipMemcpyDeviceToHost, st));
            HIP_OK(hipStreamSynchronize(st));
        }
        clock.lap(1);

        // ---- host: which candidates stay and which are pushed down ----------------------------------------------------
        // Octree.cpp:172-181: the library's partial sort, ordering on the overlap count only, on the candidates in the order
        // the reference met them; the first max_per_node stay, the rest go to the children after everything else.
        pushes.clear(); adds.clear(); kept_counts.clear();
        for (size_t j = 0; j < cand_host.size();) {
            const uint32_t n = cand_host[j].node;
            cands.clear();
            for (; j < cand_host.size() && cand_host[j].node == n; j++) cands.push_back({(uint8_t)(cand_host[j].cls & 0xFFu), (uint8_t)((cand_host[j].cls >> 8) & 0xFu), (uint32_t)j});
            if (cands.size() > P.max_per_node)
                std::partial_sort(cands.begin(), cands.begin() + (std::ptrdiff_t)P.max_per_node, cands.end(), [](const Cand &x, const Cand &y) { return x.count > y.count; });
            const size_t keep = std::min(cands.size(), P.max_per_node);
            for (size_t i = 0; i < keep; i++) L.kept.push_back({(uint32_t)i, n, cand_host[cands[i].slot].tri});
            if (keep) kept_counts.push_back(make_uint2(n, (uint32_t)keep));
            uint32_t fill[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (size_t i = P.max_per_node; i < cands.size(); i++)
                for (uint32_t o = 0; o < 8; o++)
                    if (cands[i].mask & (1u << o)) pushes.push_back({fill[o]++, n * 8u + o, cand_host[cands[i].slot].tri});
            for (uint32_t o = 0; o < 8; o++) if (fill[o]) adds.push_back(make_uint2(n * 8u + o, fill[o]));
        }
        clock.lap(2);

        // ---- the next level -------------------------------------------------------------------------------------------
        if (!kept_counts.empty()) {
            HIP_OK(kept_d.alloc(kept_counts.size() * sizeof(uint2)));
            HIP_OK(hipMemcpyAsync(kept_d.p, kept_counts.data(), kept_counts.size() * sizeof(uint2), hipMemcpyHostToDevice, st));
            kept_counts_kernel<<<grid_for(kept_counts.size()), BLOCK, 0, st>>>(kept_d.as<uint2>(), (uint32_t)kept_counts.size(), L.own.as<uint32_t>());
            HIP_OK(hipGetLastError());
        }
        if (!adds.empty()) {
            HIP_OK(add_d.alloc(adds.size() * sizeof(uint2)));
            HIP_OK(hipMemcpyAsync(add_d.p, adds.data(), adds.size() * sizeof(uint2), hipMemcpyHostToDevice, st));
            add_pushed_kernel<<<grid_for(adds.size()), BLOCK, 0, st>>>(add_d.as<uint2>(), (uint32_t)adds.size(), sizes.as<uint32_t>());
            HIP_OK(hipGetLastError());
        }
        HIP_OK(place.alloc((slots + 1) * sizeof(Place)));
        {
            hipcub::TransformInputIterator<Place, SizeToPlace, const uint32_t *> in(sizes.as<uint32_t>(), SizeToPlace());
            size_t bytes = 0;
            HIP_OK(hipcub::DeviceScan::ExclusiveScan(nullptr, bytes, in, place.as<Place>(), AddPlace(), Place{0u, 0u}, (int)(slots + 1), st));
            HIP_OK(scan_tmp.alloc(bytes));
            HIP_OK(hipcub::DeviceScan::ExclusiveScan(scan_tmp.p, bytes, in, place.as<Place>(), AddPlace(), Place{0u, 0u}, (int)(slots + 1), st));
        }
        Place total{0u, 0u};
        HIP_OK(hipMemcpyAsync(&total, place.as<Place>() + slots, sizeof total, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));      // also: the staging vectors above may be reused now
        HIP_OK(L.kid.alloc(slots * 4));
        if (total.index == 0) {                // nothing below this level
            HIP_OK(hipMemsetAsync(L.kid.p, 0xFF, slots * 4, st));
            clock.lap(3);
            break;
        }
        if (total.overflow) return rt_internal_set_error(RT_ELIMIT, "a tree level holds too many triangle references for the device builder");
        auto N = std::make_unique<Level>();
        N->n_nodes = total.index; N->n_pairs = total.first_pair;
        HIP_OK(N->pairs.alloc((size_t)N->n_pairs * sizeof(Tri)));
        HIP_OK(N->node_of.alloc((size_t)N->n_pairs * 4));
        HIP_OK(N->seg.alloc(((size_t)N->n_nodes + 1) * 4));
        HIP_OK(next_boxes.alloc((size_t)N->n_nodes * sizeof(Box)));
        next_level_kernel<<<grid_for(slots + 1), BLOCK, 0, st>>>(boxes.as<Box>(), sizes.as<uint32_t>(), place.as<Place>(), n_nodes, L.kid.as<uint32_t>(), N->seg.as<uint32_t>(), next_boxes.as<Box>());
        HIP_OK(hipGetLastError());
        scatter_kernel<<<grid_for(n_pairs), BLOCK, 0, st>>>(L.pairs.as<Tri>(), L.node_of.as<uint32_t>(), cls.as<uint16_t>(), rank.as<Count8>(), L.seg.as<uint32_t>(),
                                                             place.as<Place>(), n_pairs, N->pairs.as<Tri>(), N->node_of.as<uint32_t>());
        HIP_OK(hipGetLastError());
        if (!pushes.empty()) {
            HIP_OK(push_d.alloc(pushes.size() * sizeof(PushRec)));
            HIP_OK(hipMemcpyAsync(push_d.p, pushes.data(), pushes.size() * sizeof(PushRec), hipMemcpyHostToDevice, st));
            tail_kernel<<<grid_for(pushes.size()), BLOCK, 0, st>>>(push_d.as<PushRec>(), (uint32_t)pushes.size(), down.as<uint32_t>(), place.as<Place>(), N->pairs.as<Tri>(), N->node_of.as<uint32_t>());
            HIP_OK(hipGetLastError());
            HIP_OK(hipStreamSynchronize(st));  // `pushes` is refilled by the next level
        }
        std::swap(boxes.p, next_boxes.p); std::swap(boxes.cap, next_boxes.cap);
        levels.push_back(std::move(N));
        clock.lap(3);
    }

    // ---- flatten: sizes of the subtrees bottom-up, places top-down --------------------------------------------------------
    DevBuf overflow_d;
    HIP_OK(overflow_d.alloc(4));
    HIP_OK(hipMemsetAsync(overflow_d.p, 0, 4, st));
    for (size_t l = levels.size(); l-- > 0;) {
        Level &L = *levels[l];
        HIP_OK(L.sub.alloc((size_t)L.n_nodes * sizeof(uint2)));
        HIP_OK(L.at.alloc((size_t)L.n_nodes * sizeof(uint2)));
        const bool last = l + 1 == levels.size();
        subtree_kernel<<<grid_for(L.n_nodes), BLOCK, 0, st>>>(L.own.as<uint32_t>(), last ? nullptr : L.kid.as<uint32_t>(), last ? nullptr : levels[l + 1]->sub.as<uint2>(), L.n_nodes,
                                                               L.sub.as<uint2>(), overflow_d.as<uint32_t>());
        HIP_OK(hipGetLastError());
    }
    uint2 total{0u, 0u};
    uint32_t overflow = 0;
    HIP_OK(hipMemcpyAsync(&total, levels[0]->sub.p, sizeof total, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(&overflow, overflow_d.p, 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemsetAsync(levels[0]->at.p, 0, sizeof(uint2), st));      // the root starts both arrays
    HIP_OK(hipStreamSynchronize(st));
    if (overflow) return rt_internal_set_error(RT_ELIMIT, "the flattened octree does not fit 32-bit offsets");
    const size_t n_words = total.x, n_ref = total.y;
    DevBuf words_d, refs_d;
    HIP_OK(words_d.alloc(n_words * 4));
    HIP_OK(refs_d.alloc(n_ref * sizeof(Tri)));
    for (size_t l = 0; l < levels.size(); l++) {
        Level &L = *levels[l];
        const bool last = l + 1 == levels.size();
        place_kernel<<<grid_for(L.n_nodes), BLOCK, 0, st>>>(L.own.as<uint32_t>(), last ? nullptr : L.kid.as<uint32_t>(), last ? nullptr : levels[l + 1]->sub.as<uint2>(), L.at.as<uint2>(), L.n_nodes,
                                                             last ? nullptr : levels[l + 1]->at.as<uint2>(), words_d.as<uint32_t>());
        HIP_OK(hipGetLastError());
        if (L.n_pairs) {
            emit_kernel<<<grid_for(L.n_pairs), BLOCK, 0, st>>>(L.pairs.as<Tri>(), L.node_of.as<uint32_t>(), L.seg.as<uint32_t>(), L.at.as<uint2>(), L.n_pairs, max_leaf,
                                                                L.depth_allows_split ? 1u : 0u, refs_d.as<Tri>());
            HIP_OK(hipGetLastError());
        }
        if (!L.kept.empty()) {
            HIP_OK(kept_d.alloc(L.kept.size() * sizeof(PushRec)));
            HIP_OK(hipMemcpyAsync(kept_d.p, L.kept.data(), L.kept.size() * sizeof(PushRec), hipMemcpyHostToDevice, st));
            emit_kept_kernel<<<grid_for(L.kept.size()), BLOCK, 0, st>>>(kept_d.as<PushRec>(), (uint32_t)L.kept.size(), L.at.as<uint2>(), refs_d.as<Tri>());
            HIP_OK(hipGetLastError());
            HIP_OK(hipStreamSynchronize(st));      // kept_d is reused by the next level
        }
    }
    clock.lap(4);

    memset(out, 0, sizeof *out);
    if (scene) {
        DevBuf d_norms;
        HIP_OK(d_norms.alloc(n_vert * 16));
        if (n_vert) HIP_OK(hipMemcpyAsync(d_norms.p, norms, n_vert * 16, hipMemcpyHostToDevice, st));
        HIP_OK(hipStreamSynchronize(st));
        out->n_words = n_words; out->n_ref = n_ref; out->max_depth = max_depth;
        for (int k = 0; k < 3; k++) { out->bounds_min[k] = root_lo[k]; out->bounds_max[k] = root_hi[k]; }
        const int rc = rt_scene_create_from(d_verts.as<float>(), d_norms.as<float>(), n_vert, refs_d.as<uint32_t>(), n_ref, words_d.as<uint32_t>(), n_words, root_lo, root_hi, 1, scene);
        clock.lap(5);
        clock.report();
        return rc;
    }
    out->nodes = (uint32_t *)malloc(std::max<size_t>(1, n_words) * 4);
    out->tri_refs = (uint32_t *)malloc(std::max<size_t>(1, n_ref) * 16);
    auto fail_free = [&](int rc) { free(out->nodes); free(out->tri_refs); memset(out, 0, sizeof *out); return rc; };
    if (!out->nodes || !out->tri_refs) return fail_free(rt_internal_set_error(RT_ENOMEM, "out of memory"));
    if (hipMemcpyAsync(out->nodes, words_d.p, n_words * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
        (n_ref && hipMemcpyAsync(out->tri_refs, refs_d.p, n_ref * sizeof(Tri), hipMemcpyDeviceToHost, st) != hipSuccess) || hipStreamSynchronize(st) != hipSuccess)
        return fail_free(rt_internal_set_error(RT_ENODEVICE, "copy of the flattened octree failed"));
    out->n_words = n_words; out->n_ref = n_ref; out->max_depth = max_depth;
    for (int k = 0; k < 3; k++) { out->bounds_min[k] = root_lo[k]; out->bounds_max[k] = root_hi[k]; }
    clock.lap(5);
    clock.report();
    return RT_OK;
}

}  // namespace

extern "C" int rt_octree_build_device(const rt_mesh *meshes, size_t n_meshes, size_t depth_limit, size_t max_triangles_per_node, size_t max_triangles_per_leaf,
                                      size_t keep_triangles_after_this_many_overlaps, int device, rt_octree *out) {
    if (!out || (n_meshes && !meshes)) return rt_internal_set_error(RT_EINVAL, "null argument");
    if (depth_limit > RT_MAX_TREE_DEPTH) return rt_internal_set_error(RT_ELIMIT, "depth_limit beyond RT_MAX_TREE_DEPTH");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return rt_internal_set_error(RT_ENODEVICE, "no HIP device: the device octree builder has no CPU fallback (rt_octree_build is the host builder)");
    if (device < 0 || device >= n_dev) return rt_internal_set_error(RT_EINVAL, "device index out of range");
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(device) != hipSuccess) return rt_internal_set_error(RT_ENODEVICE, "cannot select the HIP device");
    float *verts = nullptr, *norms = nullptr;
    size_t nv = 0;
    std::vector<rt_host::Tri> all;
    int rc = rt_host::merge_meshes(meshes, n_meshes, &verts, &norms, &nv, all);
    if (rc == RT_OK) {
        const rt_host::Params P{depth_limit, max_triangles_per_node, max_triangles_per_leaf, keep_triangles_after_this_many_overlaps};
        rc = build_on_device(verts, norms, nv, all, P, out);
        if (rc == RT_OK) { out->vertices = verts; out->normals = norms; out->n_vert = nv; }
        else { free(verts); free(norms); }
    }
    (void)hipSetDevice(prev);
    return rc;
}

extern "C" int rt_scene_create_from_meshes(const rt_mesh *meshes, size_t n_meshes, size_t depth_limit, size_t max_triangles_per_node, size_t max_triangles_per_leaf,
                                           size_t keep_triangles_after_this_many_overlaps, int device, float bounds_min[3], float bounds_max[3], rt_scene **out) {
    if (!out || (n_meshes && !meshes)) return rt_internal_set_error(RT_EINVAL, "null argument");
    *out = nullptr;
    if (depth_limit > RT_MAX_TREE_DEPTH) return rt_internal_set_error(RT_ELIMIT, "depth_limit beyond RT_MAX_TREE_DEPTH");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return rt_internal_set_error(RT_ENODEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= n_dev) return rt_internal_set_error(RT_EINVAL, "device index out of range");
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(device) != hipSuccess) return rt_internal_set_error(RT_ENODEVICE, "cannot select the HIP device");
    float *verts = nullptr, *norms = nullptr;
    size_t nv = 0;
    std::vector<rt_host::Tri> all;
    int rc = rt_host::merge_meshes(meshes, n_meshes, &verts, &norms, &nv, all);
    // what rt_scene_create checks on the host arrays it is given (the device arrays the scene is made from here are not read back)
    for (size_t i = 0; rc == RT_OK && i < all.size(); i++) {
        if (all[i].a >= nv || all[i].b >= nv || all[i].c >= nv) rc = rt_internal_set_error(RT_ESCENE, "a triangle indexes a vertex that does not exist");
        else if (all[i].material >= RT_NUM_MATERIALS) rc = rt_internal_set_error(RT_ESCENE, "a triangle's material id is out of range");
    }
    if (rc == RT_OK) {
        const rt_host::Params P{depth_limit, max_triangles_per_node, max_triangles_per_leaf, keep_triangles_after_this_many_overlaps};
        rt_octree info;
        const int chosen = rt_internal_swap_device(device);      // the scene lives on the device that built it
        rc = build_on_device(verts, norms, nv, all, P, &info, out);
        (void)rt_internal_swap_device(chosen);
        if (rc == RT_OK) {
            for (int k = 0; k < 3; k++) { if (bounds_min) bounds_min[k] = info.bounds_min[k]; if (bounds_max) bounds_max[k] = info.bounds_max[k]; }
        }
    }
    free(verts); free(norms);
    (void)hipSetDevice(prev);
    return rc;
}
