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
//   totals_kernel     per node and octant the number of such pairs
//   scatter_kernel    writes the next level's pairs; child_boxes_kernel the next level's boxes
// Triangles that overlap `keep_after` or more children are the reference's "keep" candidates (Octree.cpp:167-181): at
// most max_per_node of them stay in the node, chosen by std::partial_sort on the overlap count alone, whose order among
// equal counts is whatever the library's heap leaves.  They are few, so they are copied to the host, ordered there by
// the same library call on the same sequence, and the ones pushed down are appended to the children's lists
// (tail_kernel) exactly where the reference appends them.  The host also keeps the tree's shape (a few words per
// node) and flattens it (Octree.cpp:38-85); the leaves' references are gathered into pre-order on the device
// (emit_kernel) and come back in one copy.
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

namespace {

using rt_host::Tri;
constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr uint32_t CAND_BIT = 0x8000u;     // class word: overlap mask | count << 8 | CAND_BIT
constexpr int BLOCK = 256;

struct Box { float lo[3], hi[3]; };
struct Count8 { uint32_t c[8]; };
struct CandRec { uint32_t pair, node, cls; Tri tri; };     // a keep candidate, copied to the host
struct PushRec { uint32_t dest, node; Tri tri; };          // a candidate pushed down: place in the next level's pair list

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

__global__ __launch_bounds__(BLOCK) void totals_kernel(const Count8 *__restrict__ rank, const uint32_t *__restrict__ seg, uint32_t n_nodes, uint32_t *__restrict__ totals) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_nodes * 8u) return;
    const uint32_t node = i >> 3, o = i & 7u;
    totals[i] = rank[seg[node + 1]].c[o] - rank[seg[node]].c[o];
}

__global__ __launch_bounds__(BLOCK) void candidates_kernel(const uint32_t *__restrict__ cand_idx, uint32_t n_cand, const Tri *__restrict__ pairs,
                                                           const uint32_t *__restrict__ node_of, const uint16_t *__restrict__ cls, CandRec *__restrict__ out) {
    const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= n_cand) return;
    const uint32_t i = cand_idx[j];
    out[j] = CandRec{i, node_of[i], cls[i], pairs[i]};
}

__global__ __launch_bounds__(BLOCK) void scatter_kernel(const Tri *__restrict__ pairs, const uint32_t *__restrict__ node_of, const uint16_t *__restrict__ cls,
                                                        const Count8 *__restrict__ rank, const uint32_t *__restrict__ seg, const uint32_t *__restrict__ child_base,
                                                        const uint32_t *__restrict__ child_node, uint32_t n_pairs, Tri *__restrict__ next_pairs, uint32_t *__restrict__ next_node_of) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_pairs) return;
    const uint32_t c = cls[i];
    if ((c & CAND_BIT) || (c & 0xFFu) == 0u) return;
    const uint32_t node = node_of[i];
    const Tri t = pairs[i];
    const Count8 mine = rank[i], first = rank[seg[node]];
    for (uint32_t o = 0; o < 8; o++)
        if (c & (1u << o)) {
            const uint32_t dest = child_base[node * 8u + o] + (mine.c[o] - first.c[o]);
            next_pairs[dest] = t;
            next_node_of[dest] = child_node[node * 8u + o];
        }
}

__global__ __launch_bounds__(BLOCK) void tail_kernel(const PushRec *__restrict__ recs, uint32_t n, Tri *__restrict__ next_pairs, uint32_t *__restrict__ next_node_of) {
    const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= n) return;
    const PushRec r = recs[j];
    next_pairs[r.dest] = r.tri;
    next_node_of[r.dest] = r.node;
}

__global__ __launch_bounds__(BLOCK) void child_boxes_kernel(const Box *__restrict__ boxes, const uint32_t *__restrict__ child_node, uint32_t n_nodes, Box *__restrict__ next_boxes) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_nodes * 8u) return;
    const uint32_t child = child_node[i];
    if (child == NONE) return;
    const Box b = boxes[i >> 3];
    Box out;
    rt_host::child_box(b.lo, b.hi, (int)(i & 7u), out.lo, out.hi);
    next_boxes[child] = out;
}

// leaves: the node's pairs are its final reference list; leaf_dest[node] is where that list starts in the flattened array
__global__ __launch_bounds__(BLOCK) void emit_kernel(const Tri *__restrict__ pairs, const uint32_t *__restrict__ node_of, const uint32_t *__restrict__ seg,
                                                     const uint32_t *__restrict__ leaf_dest, uint32_t n_pairs, Tri *__restrict__ refs) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_pairs) return;
    const uint32_t node = node_of[i], dest = leaf_dest[node];
    if (dest != NONE) refs[dest + (i - seg[node])] = pairs[i];
}

// ---- host side ------------------------------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p) { o.p = nullptr; }
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { if (p) { (void)hipFree(p); p = nullptr; } return hipMalloc(&p, bytes ? bytes : 1); }
    template <class T> T *as() const { return (T *)p; }
};

struct Level {                       // one tree level: host-side shape + the device arrays the final gather reads
    uint32_t n_nodes = 0, n_pairs = 0;
    std::vector<uint32_t> seg;       // [n_nodes + 1] first pair of each node
    std::vector<uint8_t> split;      // node was subdivided
    std::vector<uint32_t> kid;       // [n_nodes * 8] index in the next level, NONE for an empty child
    std::vector<uint32_t> kept_off;  // [n_nodes + 1] into `kept` (split nodes: the triangles that stay in the node)
    std::vector<Tri> kept;
    DevBuf pairs, node_of, seg_d;
};

inline uint32_t grid_for(size_t n) { return (uint32_t)((n + BLOCK - 1) / BLOCK); }

#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { char m_[256]; snprintf(m_, sizeof m_, "%s: %s", #call, hipGetErrorString(e_)); return rt_internal_set_error(RT_ENODEVICE, m_); } } while (0)

struct Cand { uint8_t mask, count; uint32_t slot; };   // slot: index into the level's candidate records

struct PhaseClock {          // RT_BUILD_TRACE=1: wall time per phase on stderr
    bool on = getenv("RT_BUILD_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    void lap(int k) { if (!on) return; auto n = std::chrono::steady_clock::now(); acc[k] += std::chrono::duration<double, std::milli>(n - t).count(); t = n; }
    void report() { if (on) fprintf(stderr, "device octree build ms: setup %.1f | kernels+readback %.1f | host ordering %.1f | scatter %.1f | flatten %.1f | emit+copy %.1f\n", acc[0], acc[1], acc[2], acc[3], acc[4], acc[5]); }
};

int build_on_device(const float *verts, size_t n_vert, const std::vector<Tri> &all, const rt_host::Params &P, rt_octree *out) {
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
        L->seg = {0u, L->n_pairs};
        HIP_OK(L->pairs.alloc((size_t)L->n_pairs * sizeof(Tri)));
        HIP_OK(L->node_of.alloc((size_t)L->n_pairs * 4));
        if (L->n_pairs) {
            HIP_OK(hipMemcpyAsync(L->pairs.p, all.data(), (size_t)L->n_pairs * sizeof(Tri), hipMemcpyHostToDevice, st));
            HIP_OK(hipMemsetAsync(L->node_of.p, 0, (size_t)L->n_pairs * 4, st));
        }
        Box rb;
        for (int k = 0; k < 3; k++) { rb.lo[k] = root_lo[k]; rb.hi[k] = root_hi[k]; }
        HIP_OK(boxes.alloc(sizeof(Box)));
        HIP_OK(hipMemcpyAsync(boxes.p, &rb, sizeof rb, hipMemcpyHostToDevice, st));
        HIP_OK(hipStreamSynchronize(st));      // rb is a stack object
        levels.push_back(std::move(L));
    }

    clock.lap(0);
    size_t max_depth = 0;
    DevBuf cls, rank, totals_d, cand_idx, cand_n, cand_recs, scan_tmp, child_base_d, child_node_d, push_d;
    std::vector<uint32_t> totals, child_base, child_node;
    std::vector<CandRec> cand_host;
    std::vector<PushRec> pushes;
    std::vector<Cand> cands;
    for (uint32_t depth = 0;; depth++) {
        Level &L = *levels.back();
        const uint32_t n_nodes = L.n_nodes, n_pairs = L.n_pairs;
        // Octree.cpp:119-125: m_max_depth is raised for every visited node below the depth limit
        const bool depth_allows_split = (size_t)depth + 1 < P.depth_limit;
        if (depth_allows_split) max_depth = std::max(max_depth, (size_t)depth + 1);
        L.split.assign(n_nodes, 0);
        L.kid.assign((size_t)n_nodes * 8, NONE);
        L.kept_off.assign((size_t)n_nodes + 1, 0);
        HIP_OK(L.seg_d.alloc(((size_t)n_nodes + 1) * 4));
        HIP_OK(hipMemcpyAsync(L.seg_d.p, L.seg.data(), ((size_t)n_nodes + 1) * 4, hipMemcpyHostToDevice, st));
        bool any_split = false;
        for (uint32_t n = 0; n < n_nodes; n++) {
            L.split[n] = depth_allows_split && (size_t)(L.seg[n + 1] - L.seg[n]) > P.max_per_leaf;
            any_split = any_split || L.split[n];
        }
        if (!any_split) { HIP_OK(hipStreamSynchronize(st)); break; }

        // ---- classify, rank, count --------------------------------------------------------------------------------
        HIP_OK(cls.alloc(((size_t)n_pairs + 1) * 2));
        HIP_OK(hipMemsetAsync((char *)cls.p + (size_t)n_pairs * 2, 0, 2, st));     // the sum runs over n_pairs + 1 entries
        const uint32_t keep_after = (uint32_t)std::min<size_t>(P.keep_after, 9), max_leaf = (uint32_t)std::min<size_t>(P.max_per_leaf, 0xFFFFFFFFu);
        classify_kernel<<<grid_for(n_pairs), BLOCK, 0, st>>>(L.pairs.as<Tri>(), L.node_of.as<uint32_t>(), boxes.as<Box>(), L.seg_d.as<uint32_t>(), d_verts.as<float4>(),
                                                              n_pairs, keep_after, max_leaf, depth_allows_split ? 1u : 0u, cls.as<uint16_t>());
        HIP_OK(hipGetLastError());
        HIP_OK(rank.alloc(((size_t)n_pairs + 1) * sizeof(Count8)));
        {
            hipcub::TransformInputIterator<Count8, GoesDown, const uint16_t *> in(cls.as<uint16_t>(), GoesDown());
            size_t bytes = 0;
            HIP_OK(hipcub::DeviceScan::ExclusiveScan(nullptr, bytes, in, rank.as<Count8>(), AddCount8(), Count8{}, (int)(n_pairs + 1), st));
            HIP_OK(scan_tmp.alloc(bytes));
            HIP_OK(hipcub::DeviceScan::ExclusiveScan(scan_tmp.p, bytes, in, rank.as<Count8>(), AddCount8(), Count8{}, (int)(n_pairs + 1), st));
        }
        HIP_OK(totals_d.alloc((size_t)n_nodes * 32));
        totals_kernel<<<grid_for((size_t)n_nodes * 8), BLOCK, 0, st>>>(rank.as<Count8>(), L.seg_d.as<uint32_t>(), n_nodes, totals_d.as<uint32_t>());
        HIP_OK(hipGetLastError());
        // ---- the keep candidates, in pair order, to the host ---------------------------------------------------------
        HIP_OK(cand_idx.alloc((size_t)n_pairs * 4));
        HIP_OK(cand_n.alloc(4));
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
        totals.resize((size_t)n_nodes * 8);
        HIP_OK(hipMemcpyAsync(totals.data(), totals_d.p, (size_t)n_nodes * 32, hipMemcpyDeviceToHost, st));
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
        // ---- host: who stays, who is pushed down, and the shape of the next level -----------------------------------
        // Octree.cpp:172-181: the library's partial sort, ordering on the overlap count only, on the candidates in the order
        // the reference met them; the first max_per_node stay, the rest go to the children after everything else.
        std::vector<uint32_t> pushed_count((size_t)n_nodes * 8, 0);
        struct Pushed { uint32_t node, slot; };
        std::vector<Pushed> pushed_order;
        L.kept.clear();
        {
            size_t j = 0;
            for (uint32_t n = 0; n < n_nodes; n++) {
                L.kept_off[n] = (uint32_t)L.kept.size();
                cands.clear();
                while (j < cand_host.size() && cand_host[j].node == n) { cands.push_back({(uint8_t)(cand_host[j].cls & 0xFFu), (uint8_t)((cand_host[j].cls >> 8) & 0xFu), (uint32_t)j}); j++; }
                if (cands.empty()) continue;
                if (cands.size() > P.max_per_node)
                    std::partial_sort(cands.begin(), cands.begin() + (std::ptrdiff_t)P.max_per_node, cands.end(), [](const Cand &x, const Cand &y) { return x.count > y.count; });
                const size_t keep = std::min(cands.size(), P.max_per_node);
                for (size_t i = 0; i < keep; i++) L.kept.push_back(cand_host[cands[i].slot].tri);
                for (size_t i = P.max_per_node; i < cands.size(); i++) {
                    pushed_order.push_back({n, cands[i].slot});
                    for (int o = 0; o < 8; o++) if (cands[i].mask & (1u << o)) pushed_count[(size_t)n * 8 + o]++;
                }
            }
            L.kept_off[n_nodes] = (uint32_t)L.kept.size();
        }
        auto N = std::make_unique<Level>();
        child_base.assign((size_t)n_nodes * 8, 0);
        child_node.assign((size_t)n_nodes * 8, NONE);
        N->seg.push_back(0u);
        uint64_t next_pairs = 0;
        for (uint32_t n = 0; n < n_nodes; n++) {
            if (!L.split[n]) continue;
            for (int o = 0; o < 8; o++) {
                const size_t q = (size_t)n * 8 + o;
                const uint32_t size = totals[q] + pushed_count[q];
                if (size == 0) continue;                 // Octree.cpp:194-198: empty children are never visited (and are dropped when flattening)
                child_base[q] = (uint32_t)next_pairs;
                child_node[q] = N->n_nodes;
                L.kid[q] = N->n_nodes;
                N->n_nodes++;
                next_pairs += size;
                if (next_pairs >= 0x7FFFFFFFull) return rt_internal_set_error(RT_ELIMIT, "a tree level holds too many triangle references for the device builder");
                N->seg.push_back((uint32_t)next_pairs);
            }
        }
        N->n_pairs = (uint32_t)next_pairs;
        pushes.clear();
        {
            std::vector<uint32_t> tail_fill((size_t)n_nodes * 8, 0);
            for (const Pushed &pd : pushed_order) {
                const CandRec &c = cand_host[pd.slot];
                for (int o = 0; o < 8; o++)
                    if (c.cls & (1u << o)) {
                        const size_t q = (size_t)pd.node * 8 + o;
                        pushes.push_back({child_base[q] + totals[q] + tail_fill[q]++, child_node[q], c.tri});
                    }
            }
        }
        if (N->n_nodes == 0) { HIP_OK(hipStreamSynchronize(st)); break; }

        clock.lap(2);
        // ---- write the next level ---------------------------------------------------------------------------------------
        HIP_OK(N->pairs.alloc((size_t)N->n_pairs * sizeof(Tri)));
        HIP_OK(N->node_of.alloc((size_t)N->n_pairs * 4));
        HIP_OK(next_boxes.alloc((size_t)N->n_nodes * sizeof(Box)));
        HIP_OK(child_base_d.alloc((size_t)n_nodes * 32));
        HIP_OK(child_node_d.alloc((size_t)n_nodes * 32));
        HIP_OK(hipMemcpyAsync(child_base_d.p, child_base.data(), (size_t)n_nodes * 32, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(child_node_d.p, child_node.data(), (size_t)n_nodes * 32, hipMemcpyHostToDevice, st));
        scatter_kernel<<<grid_for(n_pairs), BLOCK, 0, st>>>(L.pairs.as<Tri>(), L.node_of.as<uint32_t>(), cls.as<uint16_t>(), rank.as<Count8>(), L.seg_d.as<uint32_t>(),
                                                             child_base_d.as<uint32_t>(), child_node_d.as<uint32_t>(), n_pairs, N->pairs.as<Tri>(), N->node_of.as<uint32_t>());
        HIP_OK(hipGetLastError());
        if (!pushes.empty()) {
            HIP_OK(push_d.alloc(pushes.size() * sizeof(PushRec)));
            HIP_OK(hipMemcpyAsync(push_d.p, pushes.data(), pushes.size() * sizeof(PushRec), hipMemcpyHostToDevice, st));
            tail_kernel<<<grid_for(pushes.size()), BLOCK, 0, st>>>(push_d.as<PushRec>(), (uint32_t)pushes.size(), N->pairs.as<Tri>(), N->node_of.as<uint32_t>());
            HIP_OK(hipGetLastError());
        }
        child_boxes_kernel<<<grid_for((size_t)n_nodes * 8), BLOCK, 0, st>>>(boxes.as<Box>(), child_node_d.as<uint32_t>(), n_nodes, next_boxes.as<Box>());
        HIP_OK(hipGetLastError());
        HIP_OK(hipStreamSynchronize(st));      // the staging vectors above are reused by the next level
        std::swap(boxes.p, next_boxes.p);
        clock.lap(3);
        levels.push_back(std::move(N));
    }

    // ---- flatten (Octree.cpp:38-85): pre-order, children in ascending octant order ---------------------------------------
    std::vector<uint32_t> words;
    std::vector<std::vector<uint32_t>> leaf_dest(levels.size());
    for (size_t l = 0; l < levels.size(); l++) leaf_dest[l].assign(levels[l]->n_nodes, NONE);
    struct KeptCopy { uint32_t level, node, dest; };
    std::vector<KeptCopy> kept_copies;
    uint64_t n_ref = 0;
    struct Visit { uint32_t level, node; size_t parent_slot; };
    std::vector<Visit> stack = {{0u, 0u, 0u}};
    while (!stack.empty()) {
        const Visit v = stack.back();
        stack.pop_back();
        const Level &L = *levels[v.level];
        if (v.parent_slot != 0) words[v.parent_slot] = (uint32_t)words.size();
        const bool split = !L.split.empty() && L.split[v.node];
        uint32_t mask = 0, count = 0, n_tris;
        if (split) {
            for (int o = 0; o < 8; o++) if (L.kid[(size_t)v.node * 8 + o] != NONE) { mask |= 1u << o; count++; }
            n_tris = L.kept_off[v.node + 1] - L.kept_off[v.node];
            if (n_tris) kept_copies.push_back({v.level, v.node, (uint32_t)n_ref});
        } else {
            n_tris = L.seg[v.node + 1] - L.seg[v.node];
            if (n_tris) leaf_dest[v.level][v.node] = (uint32_t)n_ref;
        }
        words.push_back(((uint32_t)(uint16_t)n_tris << 16) | (mask << 8) | count);   // 16-bit count field (Octree.cpp:44)
        words.push_back((uint32_t)n_ref);
        n_ref += n_tris;
        if (n_ref > 0xFFFFFFFFull) return rt_internal_set_error(RT_ELIMIT, "more than 2^32 triangle references");
        const size_t first = words.size();
        for (uint32_t i = 0; i < count; i++) words.push_back(0u);
        size_t slot = first + count;
        for (int o = 7; o >= 0; o--) if (mask & (1u << o)) stack.push_back({v.level + 1, L.kid[(size_t)v.node * 8 + o], --slot});
    }

    clock.lap(4);
    memset(out, 0, sizeof *out);
    out->nodes = (uint32_t *)malloc(std::max<size_t>(1, words.size()) * 4);
    out->tri_refs = (uint32_t *)malloc(std::max<size_t>(1, (size_t)n_ref) * 16);
    if (!out->nodes || !out->tri_refs) { free(out->nodes); free(out->tri_refs); memset(out, 0, sizeof *out); return rt_internal_set_error(RT_ENOMEM, "out of memory"); }
    auto fail_free = [&](int rc) { free(out->nodes); free(out->tri_refs); memset(out, 0, sizeof *out); return rc; };
    if (n_ref) {
        DevBuf refs_d, dest_d;
        if (refs_d.alloc((size_t)n_ref * sizeof(Tri)) != hipSuccess) return fail_free(rt_internal_set_error(RT_ENOMEM, "out of device memory"));
        for (size_t l = 0; l < levels.size(); l++) {
            Level &L = *levels[l];
            if (!L.n_pairs) continue;
            if (dest_d.alloc((size_t)L.n_nodes * 4) != hipSuccess || !L.seg_d.p) return fail_free(rt_internal_set_error(RT_ENOMEM, "out of device memory"));
            if (hipMemcpyAsync(dest_d.p, leaf_dest[l].data(), (size_t)L.n_nodes * 4, hipMemcpyHostToDevice, st) != hipSuccess) return fail_free(rt_internal_set_error(RT_ENODEVICE, "copy failed"));
            emit_kernel<<<grid_for(L.n_pairs), BLOCK, 0, st>>>(L.pairs.as<Tri>(), L.node_of.as<uint32_t>(), L.seg_d.as<uint32_t>(), dest_d.as<uint32_t>(), L.n_pairs, refs_d.as<Tri>());
            if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail_free(rt_internal_set_error(RT_ENODEVICE, "emit kernel failed"));
        }
        if (hipMemcpyAsync(out->tri_refs, refs_d.p, (size_t)n_ref * sizeof(Tri), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return fail_free(rt_internal_set_error(RT_ENODEVICE, "copy of the reference list failed"));
        for (const KeptCopy &k : kept_copies) {
            const Level &L = *levels[k.level];
            memcpy(out->tri_refs + 4 * (size_t)k.dest, L.kept.data() + L.kept_off[k.node], (size_t)(L.kept_off[k.node + 1] - L.kept_off[k.node]) * sizeof(Tri));
        }
    }
    if (!words.empty()) memcpy(out->nodes, words.data(), words.size() * 4);
    out->n_words = words.size(); out->n_ref = (size_t)n_ref; out->max_depth = max_depth;
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
        rc = build_on_device(verts, nv, all, P, out);
        if (rc == RT_OK) { out->vertices = verts; out->normals = norms; out->n_vert = nv; }
        else { free(verts); free(norms); }
    }
    (void)hipSetDevice(prev);
    return rc;
}
