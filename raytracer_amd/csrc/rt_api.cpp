// Host side of the C ABI declared in include/rt_hotpath.h: scene validation and re-flattening, device memory,
// the render entry point.  Replaces the GL resource handling around the reference's render pass
// (app/core/src/App.cpp:23-26,96-140, Buffer.cpp:3-14, Skybox.cpp:46-57, Framebuffer.cpp:6-12).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "rt_internal.h"

namespace {

thread_local char g_err[512] = "";   // (a fixed buffer: reporting an error allocates nothing)
thread_local int g_device = -1;   // -1: not chosen, use the current HIP device

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? RT_ENOMEM : RT_ENODEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// rt_render_noalloc: the scratch of the one-kernel pipeline's cost-ordered start -- a cost per 16x16 tile measured by the
// last launch, and the tile order sorted by it -- as a file-scope device array instead of a scene-owned allocation.  Each
// device the library runs on holds its own copy (1 MiB) until the process ends.  The array's NAME is only a host-side
// stand-in: its device address is asked for once per device (noalloc_setup, with that device current) and kept; host code
// uses that pointer, never the name.
// ------------------------------------------------------------------------------------------------------------------
static __device__ __align__(256) uint32_t scratch[2u * RT_NOALLOC_MAX_TILES];   // [cost of tile | start order]

namespace {
constexpr int kMaxDevices = 64;
struct NoAllocState {
    uint32_t *ptr = nullptr;   // device address of this device's copy of `scratch`
    uint64_t key = 0;          // frame geometry the order held by the array belongs to (0: none)
    uint32_t age = 0;          // launches since the order was last rebuilt
};
NoAllocState g_noalloc[kMaxDevices];
std::mutex g_noalloc_mutex;

// Once per device, with that device current.
int noalloc_setup(int dev) {
    if (dev < 0 || dev >= kMaxDevices) return fail(RT_ELIMIT, "device %d is beyond the %d devices the library keeps scratch for", dev, kMaxDevices);
    std::lock_guard<std::mutex> lock(g_noalloc_mutex);
    if (g_noalloc[dev].ptr) return RT_OK;
    void *p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !p) return fail(RT_ENODEVICE, "hipGetSymbolAddress(scratch) on device %d: %s", dev, hipGetErrorString(e));
    g_noalloc[dev].ptr = (uint32_t *)p;
    return RT_OK;
}
}  // namespace

#define RT_N_CAND 5
static const int kCandPipeline[RT_N_CAND] = {RT_PIPELINE_MEGAKERNEL, RT_PIPELINE_WAVEFRONT, RT_PIPELINE_PERSISTENT, RT_PIPELINE_WAVEFRONT, RT_PIPELINE_PERSISTENT};

struct rt_scene {
    int device = 0;
    RtDeviceScene d{};
    void *d_vertices = nullptr, *d_normals = nullptr, *d_refs = nullptr, *d_nodes = nullptr, *d_tris = nullptr, *d_shade = nullptr, *d_dup = nullptr, *d_risky = nullptr, *d_planes = nullptr, *d_sky = nullptr;
    unsigned long long *d_counters = nullptr;
    bool counting = false;
    // staging for host-memory targets
    void *s_color = nullptr, *s_depth = nullptr, *s_kind = nullptr, *s_slot = nullptr, *s_t = nullptr, *s_nrays = nullptr, *s_ndcx = nullptr, *s_ndcy = nullptr;
    size_t s_ndc_cap = 0, s_target_cap = 0;   // pixels the ndc tables / the colour and debug planes have room for (grow-only)
    // Scene-owned scratch (cost maps, batch staging, wavefront workspace, tile counters) is shared by every render of the
    // scene: a call on another stream than the previous one first waits for that one's work.
    hipEvent_t last_use = nullptr;
    hipStream_t last_stream = nullptr;
    bool used = false;
    int n_cu = 256;
    // wavefront and persistent pipelines: up to RT_MAX_SUBFRAMES independent kernel chains on their own streams (interleaved
    // sub-frames of a frame, consecutive frames of a sequence), so that the sparse tail of one chain overlaps the dense
    // kernels of another
    int pipeline = RT_PIPELINE_AUTO;
    int subframes = 0;                 // sub-frames per frame (0: the library's choice)
    int frames_in_flight = 0;          // frames of a sequence traced concurrently (0: the library's choice)
    int frames_per_launch = 0;         // frames of a sequence sharing one chain of wavefront launches (0: the library's choice)
    struct Sub {
        RtWorkspace ws{};
        void *block = nullptr;
        size_t slots = 0;
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
        RtProbe probe{};
        uint32_t *pool = nullptr;   // tile counter of the persistent path kernel
        // one-lane-per-pixel kernel: per-unit cost measured by the last launch and the start order derived from it
        uint32_t *lpt_cost = nullptr, *lpt_order = nullptr;
        size_t lpt_units = 0, lpt_cap = 0;   // units of the geometry the block is used for, and units it has room for
        uint64_t lpt_key = 0;       // frame geometry the order belongs to (0: none)
        uint32_t lpt_age = 0;       // launches since the order was last rebuilt
    } sub[RT_MAX_SUBFRAMES];
    bool timing = false;
    hipEvent_t fork_ev = nullptr;
    void *stage = nullptr;      // sample staging of batched sequence launches (RtBatch) and of frames in flight
    size_t stage_elems = 0;
    // blocks that were outgrown while work that reads them may still be in flight: released with the scene (releasing them
    // inside an enqueue would block the caller until the device is idle)
    std::vector<void *> retired;
    int max_batch = 0;          // 0: automatic; 1: never batch
    int order = RT_ORDER_AUTO;
    int last_pipeline = -1, last_order = -1;   // what the last render call ran with (rt_scene_last_mode)
    // RT_PIPELINE_AUTO / RT_ORDER_AUTO: the three pipelines, and for the two wave-wide ones the reference's child order and the
    // exact near-first order, give identical pixels, so the scene measures the candidates on its own frames (event pairs around
    // a call's launches, read back without blocking at a later call) and uses the fastest for the workload at hand; the others
    // are measured again now and then.  Candidate c: pipeline kCandPipeline[c], exact near-first order iff c >= 3.
    struct Tuner {
        // single frames and sequences are different workloads for the one-kernel pipeline (a sequence is traced several
        // frames per launch): each kind of call keeps measurements of its own, so that alternating between them loses none
        struct Table {
            uint64_t key = 0;                   // frame size / recursion limit / band split the measurements belong to
            double ms[RT_N_CAND] = {-1.0, -1.0, -1.0, -1.0, -1.0};  // per-frame time of each candidate (< 0: not measured yet)
            uint32_t seen[RT_N_CAND] = {0, 0, 0, 0, 0};             // measurements taken (the first one of a candidate is a warm-up and is dropped)
            uint32_t calls = 0;
        } tab[2];
        int cur = 0;                       // table of the call at hand (0: one frame, 1: a sequence)
        int pending = -1;                  // candidate of the measurement in flight
        int pending_tab = 0;
        uint32_t pending_frames = 0;
        bool look_again = false;            // the periodic second look at the slower pipelines is due
        bool in_split = false;              // inside a long sequence that is issued as probe part + rest
        bool no_explore = false;            // this call must use the best known pipeline
        hipEvent_t a = nullptr, b = nullptr;
    } tuner;
};

namespace {

struct DeviceGuard {
    int prev = -1; bool ok = false;
    explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess) ok = true; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

void free_staging(rt_scene *s) {
    void **p[] = {&s->s_color, &s->s_depth, &s->s_kind, &s->s_slot, &s->s_t, &s->s_nrays, &s->s_ndcx, &s->s_ndcy};
    for (void **q : p) { if (*q) (void)hipFree(*q); *q = nullptr; }
    s->s_ndc_cap = s->s_target_cap = 0;
}

// Frame-space pixel rectangle (x0, y0, x1, y1 inclusive) that the scene's octree bounds project to under the frame's
// camera, padded; false when it is empty.  A scheduling hint only (which pixels to start first), so double-precision
// host arithmetic on the inverse of the shader's inverse view-projection matrix is good enough.
bool octree_screen_rect(const RtDeviceScene &d, const rt_uniforms &u, uint32_t width, uint32_t height, uint32_t rect[4]) {
    if (d.n_nodes == 0) return false;
    double a[4][8];
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { a[r][c] = u.inv_view_proj_mat[c * 4 + r]; a[r][4 + c] = r == c ? 1.0 : 0.0; }
    for (int i = 0; i < 4; i++) {   // Gauss-Jordan with partial pivoting
        int piv = i;
        for (int r = i + 1; r < 4; r++) if (std::fabs(a[r][i]) > std::fabs(a[piv][i])) piv = r;
        if (!(std::fabs(a[piv][i]) > 1e-300)) { rect[0] = 0; rect[1] = 0; rect[2] = width - 1; rect[3] = height - 1; return true; }
        if (piv != i) for (int c = 0; c < 8; c++) std::swap(a[i][c], a[piv][c]);
        const double inv = 1.0 / a[i][i];
        for (int c = 0; c < 8; c++) a[i][c] *= inv;
        for (int r = 0; r < 4; r++) if (r != i) { const double f = a[r][i]; if (f != 0.0) for (int c = 0; c < 8; c++) a[r][c] -= f * a[i][c]; }
    }
    double x0 = 1e300, y0 = 1e300, x1 = -1e300, y1 = -1e300;
    double q[8][4];
    for (int k = 0; k < 8; k++) {
        const double p[4] = {k & 1 ? d.bounds_max[0] : d.bounds_min[0], k & 2 ? d.bounds_max[1] : d.bounds_min[1], k & 4 ? d.bounds_max[2] : d.bounds_min[2], 1.0};
        for (int r = 0; r < 4; r++) q[k][r] = a[r][4] * p[0] + a[r][5] * p[1] + a[r][6] * p[2] + a[r][7] * p[3];
    }
    const double w_min = 1e-4;   // the box is clipped against the plane w = w_min in front of the eye
    auto add = [&](const double *c) {
        const double px = (c[0] / c[3] + 1.0) * 0.5 * width, py = (c[1] / c[3] + 1.0) * 0.5 * height;
        x0 = std::min(x0, px); x1 = std::max(x1, px); y0 = std::min(y0, py); y1 = std::max(y1, py);
    };
    for (int k = 0; k < 8; k++) {
        if (q[k][3] >= w_min) add(q[k]);
        for (int ax = 0; ax < 3; ax++) {   // the three edges leaving corner k towards a higher corner index
            const int j = k | (1 << ax);
            if (j == k || (q[k][3] >= w_min) == (q[j][3] >= w_min)) continue;
            const double t = (w_min - q[k][3]) / (q[j][3] - q[k][3]);
            const double c[4] = {q[k][0] + t * (q[j][0] - q[k][0]), q[k][1] + t * (q[j][1] - q[k][1]), q[k][2] + t * (q[j][2] - q[k][2]), w_min};
            add(c);
        }
    }
    if (!(x1 >= x0)) return false;   // the whole box is behind the eye
    x0 -= 8.0; y0 -= 8.0; x1 += 8.0; y1 += 8.0;
    if (!(x1 >= 0.0 && y1 >= 0.0 && x0 <= width - 1.0 && y0 <= height - 1.0)) return false;
    rect[0] = (uint32_t)std::max(x0, 0.0); rect[1] = (uint32_t)std::max(y0, 0.0);
    rect[2] = (uint32_t)std::min(x1, width - 1.0); rect[3] = (uint32_t)std::min(y1, height - 1.0);
    return true;
}

// The same rectangle in 16x16 tiles of the local rows of one band part (strips of band_rows rows, every parts-th).
void hot_tiles(const uint32_t rect[4], bool any, uint32_t band_rows, uint32_t parts, uint32_t hot[4]) {
    hot[0] = 1; hot[1] = 1; hot[2] = 0; hot[3] = 0;
    if (!any) return;
    const uint32_t lr0 = (rect[1] / band_rows / parts) * band_rows, lr1 = (rect[3] / band_rows / parts + 1u) * band_rows - 1u;
    hot[0] = rect[0] / 16u; hot[2] = rect[2] / 16u; hot[1] = lr0 / 16u; hot[3] = lr1 / 16u;
}

// One allocation carved into the workspace planes (256-byte aligned pieces).
int ensure_workspace(rt_scene *sc, rt_scene::Sub *s, size_t slots, size_t ranges) {
    if (s->slots >= slots && s->block) return RT_OK;
    // (an outgrown block is retired, not released: work of earlier calls may still read it.  Growing by at least half keeps the sum
    // of the retired blocks below twice the final one.)
    if (s->block) { sc->retired.push_back(s->block); s->block = nullptr; slots = std::max(slots, s->slots + s->slots / 2); ranges = std::max(ranges, (slots + 1023) / 1024); s->slots = 0; }
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t f4 = al(slots * 16), u1 = al(slots * 4);
    const size_t rg = al(ranges * 4);
    const size_t total = 6 * f4 + 4 * u1 + 2 * rg + al(8 * 4) + al((96 + RT_EYE_CAP) * 16);
    HIP_TRY(hipMalloc(&s->block, total));
    char *p = (char *)s->block;
    auto take = [&](size_t b) { char *q = p; p += b; return q; };
    for (int k = 0; k < 2; k++) { s->ws.pos_seed[k] = (float4 *)take(f4); s->ws.dir_pix[k] = (float4 *)take(f4); s->ws.color[k] = (float4 *)take(f4); }
    s->ws.t = (float *)take(u1); s->ws.sphere = (uint32_t *)take(u1); s->ws.tri = (uint32_t *)take(u1); s->ws.qt = (uint32_t *)take(u1);
    s->ws.live[0] = (uint32_t *)take(rg); s->ws.live[1] = (uint32_t *)take(rg);
    s->ws.ctr = (uint32_t *)take(al(8 * 4)); s->ws.prim = (float4 *)take(al((96 + RT_EYE_CAP) * 16));
    s->slots = slots;
    return RT_OK;
}

int upload(void **dst, const void *src, size_t bytes, bool from_device = false) {
    *dst = nullptr;
    if (bytes == 0) return RT_OK;
    HIP_TRY(hipMalloc(dst, bytes));
    HIP_TRY(hipMemcpy(*dst, src, bytes, from_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    return RT_OK;
}

}  // namespace

extern "C" {

const char *rt_last_error(void) { return g_err; }
#ifndef RT_BUILD_ID
#define RT_BUILD_ID "unknown"
#endif
const char *rt_build_id(void) { return RT_BUILD_ID; }
int rt_internal_set_error(int code, const char *msg) { return fail(code, "%s", msg ? msg : ""); }
int rt_scene_device(const rt_scene *s) { return s ? s->device : -1; }

// Verification hook (not part of the ABI): the kernels' short 1/x and sqrt sequences against the compiler's correctly rounded
// ones over all 2^32 inputs; mismatches[0] = count, mismatches[1] = an input that differs.
int rt_internal_math_check(int op, unsigned long long mismatches[2]) {
    if (!mismatches || op < 0 || op > 1) return fail(RT_EINVAL, "bad argument");
    hipError_t e = (hipError_t)rt_math_check(op, mismatches);
    if (e != hipSuccess) return fail(RT_ENODEVICE, "math check: %s", hipGetErrorString(e));
    return RT_OK;
}

// Verification hook (not part of the ABI): node records (empty slots of child blocks included), deepest level, triangle references,
// triangles lying in a face plane of their node.
int rt_internal_scene_info(rt_scene *s, uint32_t out[4]) {
    if (!s || !out) return fail(RT_EINVAL, "null argument");
    out[0] = s->d.n_nodes; out[1] = s->d.max_depth; out[2] = s->d.n_ref; out[3] = s->d.n_risky;
    return RT_OK;
}

// Development hook (not part of the ABI): copies the tile-counter block of sub-frame 0 to the host.
int rt_internal_read_pool(rt_scene *s, void *out, size_t bytes) {
    if (!s || !s->sub[0].pool || bytes > rt_pool_bytes()) return fail(RT_EINVAL, "no pool");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, s->sub[0].pool, bytes, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_device_count(int *count) {
    if (!count) return fail(RT_EINVAL, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(RT_ENODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return RT_OK;
}

int rt_set_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(RT_ENODEVICE, "no HIP device available");
    if (device < 0 || device >= n) return fail(RT_EINVAL, "device %d out of range [0,%d)", device, n);
    g_device = device;
    return RT_OK;
}

extern "C" int rt_internal_swap_device(int device) { const int before = g_device; g_device = device; return before; }

int rt_scene_create(const float *vertices, const float *normals, size_t n_vert, const uint32_t *tri_refs, size_t n_ref,
                    const uint32_t *nodes, size_t n_words, const float bounds_min[3], const float bounds_max[3], rt_scene **out) {
    return rt_scene_create_from(vertices, normals, n_vert, tri_refs, n_ref, nodes, n_words, bounds_min, bounds_max, 0, out);
}

// source_is_device: the four arrays are device memory on the current device, already checked by whoever sent them (a rank
// that received the scene from the root rank, rt_dist.cpp); the bounds are host memory either way.
int rt_scene_create_from(const float *vertices, const float *normals, size_t n_vert, const uint32_t *tri_refs, size_t n_ref,
                         const uint32_t *nodes, size_t n_words, const float bounds_min[3], const float bounds_max[3], int source_is_device, rt_scene **out) {
    if (!out) return fail(RT_EINVAL, "out is null");
    *out = nullptr;
    if (!bounds_min || !bounds_max) return fail(RT_EINVAL, "bounds are null");
    if ((n_vert && (!vertices || !normals)) || (n_ref && !tri_refs) || (n_words && !nodes)) return fail(RT_EINVAL, "null array with non-zero size");
    if (n_vert > 0xFFFFFFFFull || n_ref > 0xFFFFFFFFull || n_words > 0xFFFFFFFFull) return fail(RT_ELIMIT, "array sizes must fit 32 bits");
    // RT_BUILD_TRACE=1: wall time of the phases on stderr
    const bool trace = getenv("RT_BUILD_TRACE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(now() - t).count(); };
    auto t_phase = now();
    double t_check = 0.0, t_flatten = 0.0, t_upload = 0.0;
    for (size_t i = 0; i < n_ref && !source_is_device; i++) {
        const uint32_t *t = tri_refs + 4 * i;
        if (t[0] >= n_vert || t[1] >= n_vert || t[2] >= n_vert) return fail(RT_ESCENE, "triangle reference %zu indexes vertex >= %zu", i, n_vert);
        if (t[3] >= RT_NUM_MATERIALS) return fail(RT_ESCENE, "triangle reference %zu has material id %u >= %u", i, t[3], RT_NUM_MATERIALS);
    }
    t_check = ms_since(t_phase); t_phase = now();
    if (n_words == 1) return fail(RT_ESCENE, "node buffer has 1 word; a node needs at least 2");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(RT_ENODEVICE, "no HIP device available (this library has no CPU fallback)");
    int dev = g_device;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) return fail(RT_ENODEVICE, "hipGetDevice failed"); }

    rt_scene *s = new (std::nothrow) rt_scene();
    if (!s) return fail(RT_ENOMEM, "out of host memory");
    s->device = dev;
    DeviceGuard guard(dev);
    if (!guard.ok) { delete s; return fail(RT_ENODEVICE, "cannot select HIP device %d", dev); }
    int rc = RT_OK;
    void *d_words = nullptr;
    const bool dev_src = source_is_device != 0;
    if ((rc = upload(&s->d_vertices, vertices, n_vert * 16, dev_src)) || (rc = upload(&s->d_normals, normals, n_vert * 16, dev_src)) ||
        (rc = upload(&s->d_refs, tri_refs, n_ref * 16, dev_src)) || (rc = upload(&d_words, nodes, n_words * 4, dev_src))) {
        rt_scene_destroy(s);
        return rc;
    }
    t_upload = ms_since(t_phase); t_phase = now();
    // Everything the kernels traverse is derived from those arrays on the device: the node records (the words walked
    // breadth-first and checked, Octree.cpp:44-75 / frag:460-500), the triangle records and the leaves' inline triangles.
    RtNodeBuild nb{};
    rc = rt_build_node_records((const uint32_t *)d_words, n_words, n_ref, bounds_min, bounds_max, &nb);
    if (d_words) (void)hipFree(d_words);
    if (rc != RT_OK) {
        rt_scene_destroy(s);
        switch (nb.error) {
        case 1: return fail(rc, "node at word %u runs past the node buffer (%zu words)", nb.error_word, n_words);
        case 2: return fail(rc, "node at word %u: child count does not match its child mask", nb.error_word);
        case 3: return fail(rc, "node at word %u: child count field has bits above 0xF set", nb.error_word);
        case 4: return fail(rc, "node at word %u: child pointers run past the node buffer", nb.error_word);
        case 5: return fail(rc, "node at word %u: triangle range exceeds the %zu references", nb.error_word, n_ref);
        case RT_NODE_ERROR_TOO_DEEP: return fail(rc, "octree deeper than %u levels (or cyclic)", RT_MAX_TREE_DEPTH);
        case RT_NODE_ERROR_TOO_MANY: return fail(rc, "node buffer describes more than %zu nodes (shared or cyclic children)", n_words);
        case RT_NODE_ERROR_RECORD_LIMIT: return fail(rc, "octree needs more than %u node records", RT_MAX_NODE_RECORDS);
        default: return fail(rc, "node records: %s", hipGetErrorString((hipError_t)nb.hip_error));
        }
    }
    s->d_nodes = nb.nodes;
    t_flatten = ms_since(t_phase); t_phase = now();
    if (n_ref) {
        hipError_t e = hipMalloc(&s->d_tris, n_ref * sizeof(RtTriRec));
        if (e == hipSuccess) e = hipMalloc(&s->d_shade, n_ref * sizeof(RtShadeRec));
        if (e == hipSuccess) e = hipMalloc(&s->d_dup, n_ref * sizeof(uint32_t));
        if (e == hipSuccess) e = (hipError_t)rt_build_scene_records((const float *)s->d_vertices, (const float *)s->d_normals, (const uint32_t *)s->d_refs, (uint32_t)n_ref, (RtNodeRec *)s->d_nodes, nb.n_nodes,
                                                                    (RtTriRec *)s->d_tris, (RtShadeRec *)s->d_shade, (uint32_t *)s->d_dup, (RtTriRec **)&s->d_risky, &s->d.n_risky);
        if (e == hipSuccess) e = (hipError_t)rt_build_plane_records((const RtTriRec *)s->d_tris, (uint32_t)n_ref, (RtPlaneRec **)&s->d_planes, &s->d.n_planes);
        if (e != hipSuccess) { rt_scene_destroy(s); return fail(e == hipErrorOutOfMemory ? RT_ENOMEM : RT_ENODEVICE, "triangle records: %s", hipGetErrorString(e)); }
    }
    if (hipMalloc((void **)&s->d_counters, 16 * sizeof(unsigned long long)) != hipSuccess || hipMemset(s->d_counters, 0, 128) != hipSuccess) {
        rt_scene_destroy(s);
        return fail(RT_ENOMEM, "cannot allocate counters");
    }
    // the uploaded arrays have served: every kernel reads the records derived from them (node, triangle and shading records)
    for (void **q : {&s->d_vertices, &s->d_normals, &s->d_refs}) { if (*q) (void)hipFree(*q); *q = nullptr; }
    s->d.vertices = nullptr; s->d.normals = nullptr; s->d.tri_refs = nullptr;
    s->d.nodes = (const RtNodeRec *)s->d_nodes; s->d.tris = (const RtTriRec *)s->d_tris; s->d.shade = (const RtShadeRec *)s->d_shade; s->d.dup_parent = (const uint32_t *)s->d_dup; s->d.risky_tris = (const RtTriRec *)s->d_risky; s->d.planes = (const RtPlaneRec *)s->d_planes; s->d.sky = nullptr;
    s->d.n_vert = (uint32_t)n_vert; s->d.n_ref = (uint32_t)n_ref; s->d.n_nodes = nb.n_nodes;
    s->d.sky_w = s->d.sky_h = 0;
    for (int k = 0; k < 3; k++) { s->d.bounds_min[k] = bounds_min[k]; s->d.bounds_max[k] = bounds_max[k]; }
    s->d.max_depth = nb.max_depth;
    {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) s->n_cu = v;
        if (hipEventCreateWithFlags(&s->last_use, hipEventDisableTiming) != hipSuccess) { rt_scene_destroy(s); return fail(RT_ENODEVICE, "cannot create an event"); }
        if (int nrc = noalloc_setup(dev)) { rt_scene_destroy(s); return nrc; }   // (this device's scratch array of rt_render_noalloc)
    }
    *out = s;
    if (trace) fprintf(stderr, "rt_scene_create ms: reference check %.1f | upload %.1f | node records %.1f (%u) | triangle records %.1f\n", t_check, t_upload, t_flatten, nb.n_nodes, ms_since(t_phase));
    return RT_OK;
}

int rt_skybox_set(rt_scene *s, const uint8_t *const faces[6], int w, int h) { return rt_skybox_set_from(s, faces, w, h, 0); }

int rt_skybox_set_from(rt_scene *s, const uint8_t *const faces[6], int w, int h, int source_is_device) {
    if (!s || !faces) return fail(RT_EINVAL, "null argument");
    if (w <= 0 || h <= 0 || w > 16384 || h > 16384) return fail(RT_EINVAL, "skybox size %dx%d out of range", w, h);
    for (int f = 0; f < 6; f++) if (!faces[f]) return fail(RT_EINVAL, "face %d is null", f);
    DeviceGuard guard(s->device);
    if (!guard.ok) return fail(RT_ENODEVICE, "cannot select HIP device %d", s->device);
    const size_t fb = (size_t)w * h * 4;
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, fb * 6));
    for (int f = 0; f < 6; f++) {
        hipError_t e = hipMemcpy((char *)p + fb * f, faces[f], fb, source_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(p); return fail(RT_ENODEVICE, "skybox upload: %s", hipGetErrorString(e)); }
    }
    if (s->d_sky) (void)hipFree(s->d_sky);
    s->d_sky = p;
    s->d.sky = (const uint8_t *)p; s->d.sky_w = w; s->d.sky_h = h;
    return RT_OK;
}

int rt_scene_set_pipeline(rt_scene *s, int pipeline) {
    if (!s) return fail(RT_EINVAL, "scene is null");
    if (pipeline < RT_PIPELINE_MEGAKERNEL || pipeline > RT_PIPELINE_AUTO) return fail(RT_EINVAL, "unknown pipeline %d", pipeline);
    s->pipeline = pipeline;
    return RT_OK;
}

// Which candidates (see rt_scene::Tuner) the scene's settings allow: a fixed pipeline or a fixed order narrows the choice; the
// exact near-first order exists in the two wave-wide pipelines only, and not for a scene with too many face-coplanar triangles.
static uint32_t candidate_mask(const rt_scene *s) {
    uint32_t m = 0;
    const bool exact_ok = s->d.n_risky != RT_RISKY_TOO_MANY && s->d.n_nodes != 0;
    for (int c = 0; c < RT_N_CAND; c++) {
        if (s->pipeline != RT_PIPELINE_AUTO && kCandPipeline[c] != s->pipeline) continue;
        const bool exact = c >= 3;
        // with counters on, the automatic order is the reference's: the counters are defined as the work of the shader's traversal
        if (exact && (s->order != RT_ORDER_AUTO || !exact_ok || s->counting)) continue;
        m |= 1u << c;
    }
    return m;
}
// Child order a candidate runs with under the scene's setting.
static int candidate_order(const rt_scene *s, int c) {
    if (s->order == RT_ORDER_AUTO) return c >= 3 ? RT_ORDER_NEAR_FIRST_EXACT : RT_ORDER_REFERENCE;
    if (s->order == RT_ORDER_NEAR_FIRST_EXACT && (s->d.n_risky == RT_RISKY_TOO_MANY || kCandPipeline[c] == RT_PIPELINE_MEGAKERNEL)) return RT_ORDER_REFERENCE;
    return s->order;
}
static int best_candidate(const rt_scene *s, const rt_scene::Tuner::Table &tb, uint32_t mask) {
    int best = -1;
    for (int c = 0; c < RT_N_CAND; c++) if ((mask >> c & 1u) && tb.ms[c] >= 0.0 && (best < 0 || tb.ms[c] < tb.ms[best])) best = c;
    if (best < 0) {
        // Nothing measured yet (a caller that enqueues many frames without waiting gets its first measurement late): the
        // one-kernel pipeline for small scenes, the wavefront pipeline -- nearest first where that is a candidate -- for scenes
        // of a million references and more, where it measured two to three times faster on every scene tried.
        static const int large[RT_N_CAND] = {3, 1, 4, 2, 0}, small[RT_N_CAND] = {0, 2, 1, 4, 3};
        const int *pref = s->d.n_ref >= (1u << 20) ? large : small;
        for (int k = 0; k < RT_N_CAND && best < 0; k++) if (mask >> pref[k] & 1u) best = pref[k];
    }
    return best < 0 ? 0 : best;
}

int rt_scene_pipeline_state(rt_scene *s, int *selected, double ms_per_frame[3]) {
    if (!s || !selected || !ms_per_frame) return fail(RT_EINVAL, "null argument");
    const rt_scene::Tuner::Table &tb = s->tuner.tab[s->tuner.cur];   // (of the kind of call made last)
    for (int p = 0; p < 3; p++) ms_per_frame[p] = -1.0;
    for (int c = 0; c < RT_N_CAND; c++) {
        const int p = kCandPipeline[c];
        if (tb.ms[c] >= 0.0 && (ms_per_frame[p] < 0.0 || tb.ms[c] < ms_per_frame[p])) ms_per_frame[p] = tb.ms[c];
    }
    *selected = kCandPipeline[best_candidate(s, tb, candidate_mask(s))];
    return RT_OK;
}

int rt_scene_selection_state(rt_scene *s, int *pipeline, int *order, double ms_per_frame[5]) {
    if (!s || !pipeline || !order || !ms_per_frame) return fail(RT_EINVAL, "null argument");
    const rt_scene::Tuner::Table &tb = s->tuner.tab[s->tuner.cur];
    for (int c = 0; c < RT_N_CAND; c++) ms_per_frame[c] = tb.ms[c];
    const int c = best_candidate(s, tb, candidate_mask(s));
    *pipeline = kCandPipeline[c]; *order = candidate_order(s, c);
    return RT_OK;
}

int rt_scene_last_mode(rt_scene *s, int *pipeline, int *order) {
    if (!s || !pipeline || !order) return fail(RT_EINVAL, "null argument");
    if (s->last_pipeline < 0) return fail(RT_EINVAL, "the scene has not been rendered yet");
    *pipeline = s->last_pipeline; *order = s->last_order;
    return RT_OK;
}

int rt_scene_set_traversal_order(rt_scene *s, int order) {
    if (!s) return fail(RT_EINVAL, "scene is null");
    if (order != RT_ORDER_REFERENCE && order != RT_ORDER_NEAR_FIRST && order != RT_ORDER_NEAR_FIRST_EXACT && order != RT_ORDER_AUTO) return fail(RT_EINVAL, "unknown traversal order %d", order);
    s->order = order;
    return RT_OK;
}

int rt_scene_set_sequence_batch(rt_scene *s, int max_frames) {
    if (!s) return fail(RT_EINVAL, "scene is null");
    if (max_frames < 0 || max_frames > RT_MAX_BATCH) return fail(RT_EINVAL, "batch must be in [0,%d]", RT_MAX_BATCH);
    s->max_batch = max_frames;
    return RT_OK;
}

int rt_scene_set_frames_in_flight(rt_scene *s, int n) {
    if (!s) return fail(RT_EINVAL, "scene is NULL");
    if (n < 0 || n > RT_MAX_SUBFRAMES) return fail(RT_EINVAL, "frames in flight must be in [0,%d]", RT_MAX_SUBFRAMES);
    s->frames_in_flight = n;
    return RT_OK;
}
int rt_scene_set_frames_per_launch(rt_scene *s, int n) {
    if (!s) return fail(RT_EINVAL, "scene is NULL");
    if (n < 0 || n > RT_MAX_STACK) return fail(RT_EINVAL, "frames per launch must be in [0,%d]", RT_MAX_STACK);
    s->frames_per_launch = n;
    return RT_OK;
}
int rt_scene_set_subframes(rt_scene *s, int n) {
    if (!s) return fail(RT_EINVAL, "scene is null");
    if (n < 0 || n > RT_MAX_SUBFRAMES) return fail(RT_EINVAL, "subframes must be in [0,%d]", RT_MAX_SUBFRAMES);
    s->subframes = n;
    return RT_OK;
}

int rt_scene_set_counting(rt_scene *s, int enabled) {
    if (!s) return fail(RT_EINVAL, "scene is null");
    s->counting = enabled != 0;
    return RT_OK;
}

// Frames per batched launch (see RtBatch).  RT_BATCH_BLOCK / 64 = 4 frames are blended inside the kernel and need no staging
// plane, so that is the batch whenever four frames' worth of this launch's pixels fill the machine; below
// RT_FUSED_MIN_PIXELS pixels per frame (measured: one of 8 ranks' share of a 4K frame, 1.04 M pixels, 0.087 ms per frame
// in fours and 0.078 in eights; one of 4 ranks' share, 2.07 M, 0.150 against 0.153) RT_MAX_BATCH frames go through staging.
#ifndef RT_FUSED_MIN_PIXELS
#define RT_FUSED_MIN_PIXELS (((size_t)1 << 20) + ((size_t)1 << 19))
#endif
#define RT_PROBE_PAIRS 4096u
int rt_scene_set_kernel_timing(rt_scene *s, int enabled) {
    if (!s) return fail(RT_EINVAL, "scene is null");
    DeviceGuard guard(s->device);
    if (!guard.ok) return fail(RT_ENODEVICE, "cannot select HIP device %d", s->device);
    if (enabled) {
        for (rt_scene::Sub &sb : s->sub) {
            if (!sb.probe.events) {
                sb.probe.events = new hipEvent_t[2 * RT_PROBE_PAIRS]();
                sb.probe.launches = new uint32_t[RT_PROBE_PAIRS]();
                sb.probe.capacity = RT_PROBE_PAIRS;
                for (uint32_t i = 0; i < 2 * RT_PROBE_PAIRS; i++) HIP_TRY(hipEventCreate(&sb.probe.events[i]));
            }
            sb.probe.used = 0;
        }
    }
    s->timing = enabled != 0;
    return RT_OK;
}

int rt_scene_kernel_time(rt_scene *s, double *total_ms, uint32_t *launches) {
    if (!s || !total_ms || !launches) return fail(RT_EINVAL, "null argument");
    DeviceGuard guard(s->device);
    if (!guard.ok) return fail(RT_ENODEVICE, "cannot select HIP device %d", s->device);
    HIP_TRY(hipDeviceSynchronize());
    double ms = 0.0; uint32_t n = 0;
    for (rt_scene::Sub &sb : s->sub) {
        for (uint32_t i = 0; i < sb.probe.used; i++) {
            float t = 0.0f;
            HIP_TRY(hipEventElapsedTime(&t, sb.probe.events[2 * i], sb.probe.events[2 * i + 1]));
            ms += t; n += sb.probe.launches[i];
        }
        sb.probe.used = 0;
    }
    *total_ms = ms; *launches = n;
    return RT_OK;
}

int rt_scene_stats(rt_scene *s, rt_stats *out) {
    if (!s || !out) return fail(RT_EINVAL, "null argument");
    DeviceGuard guard(s->device);
    if (!guard.ok) return fail(RT_ENODEVICE, "cannot select HIP device %d", s->device);
    unsigned long long c[16];
    HIP_TRY(hipMemcpy(c, s->d_counters, sizeof c, hipMemcpyDeviceToHost));
    out->rays = c[0]; out->node_pops = c[1]; out->triangle_tests = c[2]; out->child_box_tests = c[3];
    out->triangle_hits_shaded = c[4]; out->rays_entering_octree = c[5]; out->sphere_tests = c[6];
    out->node_pops_from_lds = c[7]; out->exact_fallbacks = c[8]; out->exact_candidate_checks = c[9];
    return RT_OK;
}

static uint32_t band_rows_of(const rt_frame *f) {
    if (!f || f->band_rows == 0 || f->band_parts == 0 || f->band_part >= f->band_parts) return 0;
    const uint32_t n_strips = (f->height + f->band_rows - 1) / f->band_rows;
    const uint32_t mine = n_strips > f->band_part ? (n_strips - f->band_part + f->band_parts - 1) / f->band_parts : 0;
    return mine * f->band_rows;
}
uint32_t rt_band_row_count(const rt_frame *f) { return band_rows_of(f); }

// Automatic selection: collects a finished measurement, then picks the candidate of this call among those of `mask` and says
// whether to time it.
static int tuner_choose(rt_scene *s, uint64_t key, bool sequence, uint32_t mask, bool *measure) {
    rt_scene::Tuner &tn = s->tuner;
    *measure = false;
    int first = 0;
    while (first < RT_N_CAND - 1 && !(mask >> first & 1u)) first++;
    if (!tn.a && (hipEventCreate(&tn.a) != hipSuccess || hipEventCreate(&tn.b) != hipSuccess)) { tn.a = tn.b = nullptr; return first; }
    if (tn.pending >= 0 && hipEventQuery(tn.b) == hipSuccess) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, tn.a, tn.b) == hipSuccess && tn.pending_frames > 0) {
            const double per_frame = (double)ms / tn.pending_frames;
            const int p = tn.pending;
            rt_scene::Tuner::Table &pt = tn.tab[tn.pending_tab];
            if (pt.seen[p]++ > 0) pt.ms[p] = pt.ms[p] < 0.0 ? per_frame : 0.5 * (pt.ms[p] + per_frame);
        }
        tn.pending = -1;
    }
    tn.cur = sequence ? 1 : 0;
    rt_scene::Tuner::Table &t = tn.tab[tn.cur];
    if (key != t.key) { t.key = key; for (int p = 0; p < RT_N_CAND; p++) { t.ms[p] = -1.0; t.seen[p] = 0; } t.calls = 0; if (tn.pending_tab == tn.cur) tn.pending = -1; }
    t.calls++;
    int choice = best_candidate(s, t, mask);
    if (t.calls % 64u == 0u) tn.look_again = true;
    if (tn.pending < 0) {
        *measure = true;
        if (tn.no_explore) return choice;
        for (int p = 0; p < RT_N_CAND; p++) if ((mask >> p & 1u) && t.ms[p] < 0.0) return p;   // something still unknown: try it
        if (tn.look_again) {                                                                   // look at the others again now and then
            tn.look_again = false;
            int other = choice;
            for (uint32_t k = 0, want = 1u + (t.calls / 64u) % (RT_N_CAND - 1u); k < want; ) {
                other = (other + 1) % RT_N_CAND;
                if (other == choice) break;
                if (mask >> other & 1u) k++;
            }
            choice = other;
        }
    }
    return choice;
}

// Frames of one sequence share everything that decides which pixels a launch owns and where they live.
static bool same_geometry(const rt_frame &a, const rt_frame &b) {
    return a.width == b.width && a.height == b.height && a.band_rows == b.band_rows && a.band_parts == b.band_parts &&
           a.band_part == b.band_part && a.compact_rows == b.compact_rows && a.ndc_x == b.ndc_x && a.ndc_y == b.ndc_y && a.stream == b.stream;
}

static int render_frames(rt_scene *s, const rt_frame *frames, size_t n_frames, const rt_target *t) {
    if (!s || !frames || !t) return fail(RT_EINVAL, "null argument");
    if (n_frames == 0) return RT_OK;
    const rt_frame *f = frames;
    if (!t->color) return fail(RT_EINVAL, "target colour plane is null");
    if (f->width == 0 || f->height == 0 || f->width > 65536 || f->height > 65536) return fail(RT_EINVAL, "frame size %ux%u out of range", f->width, f->height);
    if (f->band_parts == 0 || f->band_part >= f->band_parts || f->band_rows == 0) return fail(RT_EINVAL, "band selection %u/%u (rows %u) invalid", f->band_part, f->band_parts, f->band_rows);
    if (t->memory != RT_MEM_HOST && t->memory != RT_MEM_DEVICE) return fail(RT_EINVAL, "target memory kind %d unknown", t->memory);
    if ((f->ndc_x == nullptr) != (f->ndc_y == nullptr)) return fail(RT_EINVAL, "ndc_x and ndc_y must be given together");
    if (t->memory == RT_MEM_HOST && !t->sync) return fail(RT_EINVAL, "host-memory targets require sync = 1");
    if ((unsigned long long)f->width * f->height >= 0xFFFFFFFFull) return fail(RT_ELIMIT, "frame has too many pixels");
    uint32_t min_limit = 0xFFFFFFFFu;
    for (size_t i = 0; i < n_frames; i++) {
        if (frames[i].accum_index == 0) return fail(RT_EINVAL, "accum_index must be >= 1");
        if (!same_geometry(frames[i], *f)) return fail(RT_EINVAL, "frame %zu of the sequence differs from frame 0 in size, band, ndc table or stream", i);
        if (frames[i].u.max_recursion_limit < min_limit) min_limit = frames[i].u.max_recursion_limit;
    }
    // Trying a pipeline other than the best known one must not cost a whole long sequence (64 samples of a still camera
    // at 8K are 0.2 s): a long sequence is issued as a short probe part, which the selection may use to try a pipeline,
    // and the rest, which always runs on the best one known.
    const size_t probe_frames = (size_t)f->width * f->height >= ((size_t)1 << 23) ? 4 : 8;
    const uint32_t cand_mask = candidate_mask(s);
    const bool choosing = (cand_mask & (cand_mask - 1u)) != 0u;   // more than one candidate: the scene measures and selects
    if (choosing && !s->tuner.in_split && !s->counting && t->memory == RT_MEM_DEVICE && n_frames > probe_frames) {
        rt_target head = *t;
        head.sync = 0;
        s->tuner.in_split = true;
        int rc = render_frames(s, frames, probe_frames, &head);
        s->tuner.no_explore = true;
        if (rc == RT_OK) rc = render_frames(s, frames + probe_frames, n_frames - probe_frames, t);
        s->tuner.no_explore = false;
        s->tuner.in_split = false;
        return rc;
    }
    DeviceGuard guard(s->device);
    if (!guard.ok) return fail(RT_ENODEVICE, "cannot select HIP device %d", s->device);
    hipStream_t st = (hipStream_t)f->stream;
    // the scene's scratch buffers are written by every render: work issued on another stream is ordered behind the last call's
    if (s->used && s->last_stream != st) HIP_TRY(hipStreamWaitEvent(st, s->last_use, 0));
    auto mark_use = [&]() -> int {
        if (hipEventRecord(s->last_use, st) != hipSuccess) return fail(RT_ENODEVICE, "hipEventRecord failed");
        s->last_stream = st; s->used = true;
        return RT_OK;
    };
    auto note_mode = [&](int pl, uint32_t near_first) { s->last_pipeline = pl; s->last_order = (pl == RT_PIPELINE_MEGAKERNEL && near_first == RT_ORDER_NEAR_FIRST_EXACT) ? RT_ORDER_REFERENCE : (int)near_first; };

    if (f->compact_rows > 1) return fail(RT_EINVAL, "compact_rows must be 0 or 1");
    RtLaunch L{};
    L.u = f->u;
    L.width = f->width; L.height = f->height; L.accum_index = f->accum_index; L.blend = f->blend ? 1u : 0u;
    L.band_rows = f->band_rows; L.band_parts = f->band_parts; L.band_part = f->band_part;
    L.n_local_rows = band_rows_of(f);   // rows owned by this part: whole strips, the last one clipped by the kernel
    L.compact_rows = f->compact_rows;
    L.n_cu = (uint32_t)s->n_cu;
    L.counters = s->counting ? s->d_counters : nullptr;
    if (s->counting) HIP_TRY(hipMemsetAsync(s->d_counters, 0, 128, st));

    L.sub_parts = 1; L.sub_part = 0;
    bool measure = false;
    int cand = 0;
    while (cand < RT_N_CAND - 1 && !(cand_mask >> cand & 1u)) cand++;
    if (choosing) {
        // (with the slot plane of the primary hit requested the exact order also resolves the reference's slot: another workload)
        const uint64_t key = ((uint64_t)f->width << 40) ^ ((uint64_t)f->height << 20) ^ ((uint64_t)min_limit << 8) ^ ((uint64_t)f->band_parts << 4) ^ (uint64_t)(s->counting ? 1 : 0) ^
                             ((uint64_t)s->order << 1) ^ (uint64_t)(t->hit_slot ? 8 : 0) ^ ((uint64_t)cand_mask << 56);
        cand = tuner_choose(s, key + 1, n_frames > 1, cand_mask, &measure);
    }
    int pipeline = kCandPipeline[cand];
    const int order = candidate_order(s, cand);
    L.near_first = (uint32_t)order;   // (RT_ORDER_NEAR_FIRST_EXACT in the one-kernel pipeline: that kernel runs the reference's order, see rt_hotpath.h)
    bool wavefront = pipeline == RT_PIPELINE_WAVEFRONT && min_limit >= 1 && L.n_local_rows > 0;
    bool persistent = pipeline == RT_PIPELINE_PERSISTENT && L.n_local_rows > 0;
    // a measurement belongs to the pipeline that actually runs every frame of the call: frames without any segment go
    // through the one-kernel pipeline whatever was chosen
    if (measure && (pipeline != RT_PIPELINE_MEGAKERNEL) && (min_limit < 1 || !(wavefront || persistent))) measure = false;
    // sub-frames per frame (every one should still fill the machine: >= 2^19 pixels) and frames in flight
    uint32_t K = 1, F = 1, V = 1;
    // (the one-kernel-per-frame pipeline orders its workgroups by measured cost instead, see attach_order: with that its
    // kernels end without a sparse tail and sub-frames bring it nothing)
    if (wavefront || persistent) {
        const uint32_t strips = L.n_local_rows / f->band_rows;
        // A frame is cut into K interleaved sub-frames whose kernels run on separate streams, and a sequence keeps F
        // consecutive frames in flight: chain (frame mod F, sub-frame) has a stream and a workspace of its own and traces
        // into a staging plane; the samples are blended into the target strictly in frame order (each blend waits for the
        // previous frame's blend of the same sub-frame), so every pixel sees the arithmetic of frame-after-frame
        // rendering.  Either way the sparse end of one chain's kernels overlaps dense work of the others.  Frames in
        // flight do that without shrinking the kernels and measure faster than sub-frames wherever both apply
        // (4K, five bounces, 1.28 M triangles: 13.3 ms per frame with 1 x 3 chains, 14.3 with 3 x 1, 14.0 with 2 x 3),
        // so the library's own choice is 1 x 3 for a sequence and 3 x 1 for a single frame.
        // (The staging plane is indexed like a compact target, which a full-frame target only is when the share is the
        // whole frame.)
        const bool can_fly = n_frames >= 2 && min_limit >= 1 && (f->compact_rows || f->band_parts == 1) && s->frames_in_flight != 1;
        K = s->subframes > 0 ? (uint32_t)s->subframes : (can_fly ? 1u : 3u);
        while (K > 1 && (size_t)(strips / K) * f->band_rows * f->width < ((size_t)1 << 19)) K--;
        if (can_fly) {
            F = s->frames_in_flight > 0 ? (uint32_t)s->frames_in_flight : (K == 1 ? 3u : 1u);
            F = std::max<uint32_t>(1u, std::min<uint32_t>(F, (uint32_t)std::min<size_t>(n_frames, RT_MAX_SUBFRAMES / K)));
            // Consecutive frames that differ in `time` only (the samples of a still camera) can also share their launches
            // (RtStack): V of them are traced by one chain of kernels V times as large.  Not when the caller wants the
            // other planes (their pixels are addressed through the target's own layout), and at most 2^25 slots per
            // chain, the size of an 8K frame.
            const bool planes = t->depth || t->hit_kind || t->hit_slot || t->hit_t || t->n_rays;
            if (wavefront && F > 1 && !planes) {
                rt_frame fj = *f;
                fj.band_parts = f->band_parts * K; fj.band_part = f->band_part;
                const size_t slots = rt_workspace_slots(f->width, band_rows_of(&fj));
                V = s->frames_per_launch > 0 ? (uint32_t)s->frames_per_launch : RT_MAX_STACK;
                while (V > 1 && slots * V > ((size_t)1 << 25)) V--;
                V = std::min<uint32_t>(V, (uint32_t)((n_frames + F - 1) / F));   // (every chain should get a group)
            }
        }
        for (uint32_t j = 0; j < K * F; j++) {   // chain j = (frame mod F) * K + sub-frame
            rt_frame fj = *f;
            fj.band_parts = f->band_parts * K; fj.band_part = (j % K) * f->band_parts + f->band_part;
            const uint32_t rows = band_rows_of(&fj);
            if (wavefront) {
                int wrc = ensure_workspace(s, &s->sub[j], rt_workspace_slots(f->width, rows, V), rt_workspace_ranges(f->width, rows, V));
                if (wrc != RT_OK && j >= K) {   // no memory for another frame in flight: fewer of them
                    (void)hipGetLastError();
                    F = j / K;
                    break;
                }
                if (wrc != RT_OK) {
                    // (about 112 bytes per pixel.)  Under automatic selection a pipeline that cannot get its workspace is
                    // given up for this workload instead of failing the render: the one-kernel pipeline needs none.
                    if (s->pipeline != RT_PIPELINE_AUTO) return wrc;
                    for (int c = 0; c < RT_N_CAND; c++) if (kCandPipeline[c] == RT_PIPELINE_WAVEFRONT) { s->tuner.tab[s->tuner.cur].ms[c] = 1e30; s->tuner.tab[s->tuner.cur].seen[c] = 2; }
                    (void)hipGetLastError();   // (the failed allocation's error is handled here)
                    wavefront = false; measure = false; pipeline = RT_PIPELINE_MEGAKERNEL; cand = 0; K = 1; F = 1; V = 1;
                    L.near_first = (uint32_t)candidate_order(s, 0);
                    break;
                }
            } else if (persistent && !s->sub[j].pool) {
                HIP_TRY(hipMalloc((void **)&s->sub[j].pool, rt_pool_bytes()));
            }
            if ((K > 1 || F > 1) && !s->sub[j].stream) {
                HIP_TRY(hipStreamCreateWithFlags(&s->sub[j].stream, hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&s->sub[j].done, hipEventDisableTiming));
            }
        }
        if ((K > 1 || F > 1) && !s->fork_ev) HIP_TRY(hipEventCreateWithFlags(&s->fork_ev, hipEventDisableTiming));
        if (F > 1) {
            const size_t spx = (size_t)f->width * L.n_local_rows;
            if (s->stage_elems < (size_t)F * V * spx) {
                if (s->stage) { s->retired.push_back(s->stage); s->stage = nullptr; s->stage_elems = 0; }
                if (hipMalloc(&s->stage, (size_t)F * V * spx * 16) != hipSuccess) { (void)hipGetLastError(); F = 1; V = 1; }
                else s->stage_elems = (size_t)F * V * spx;
            }
        }
    }
    // Enqueues the sequence.  Single-kernel pipeline, or one sub-frame and one frame in flight: frame after frame on the
    // caller's stream.  Otherwise the K x F chains are forked from the caller's stream once and joined back once.  Without
    // frames in flight each sub-frame chain runs its share of every frame in order on its own stream and blends into the
    // target itself (a pixel belongs to one sub-frame, so its accumulation order is kept); with them, chain (i mod F, j)
    // traces sub-frame j of frame i into staging plane i mod F and blends it after frame i-1's blend of the same sub-frame.
    // Either way the chains drift apart and the sparse end of one kernel overlaps dense work of others.
    // One-lane-per-pixel launches start their workgroups in the order of the cost (duration) that the previous launch with
    // the same frame geometry measured per workgroup: progressive accumulation renders the same view again and again, and
    // a moving camera changes the cost map only gradually.  The first launch of a geometry uses the order given by the
    // octree's screen rectangle.  Every launch leaves its costs, sorted, for the next.
    auto attach_order = [&](RtLaunch &Li, uint32_t j, const rt_uniforms &u, hipStream_t stream) -> int {
        rt_scene::Sub &sb = s->sub[j];
        const size_t units = rt_render_units(Li.width, Li.n_local_rows);
        if (units == 0) return 0;
        if (sb.lpt_units != units) {
            if (sb.lpt_cap < units) {
                if (sb.lpt_cost) s->retired.push_back(sb.lpt_cost);
                sb.lpt_cost = nullptr; sb.lpt_cap = 0; sb.lpt_units = 0;
                if (hipMalloc((void **)&sb.lpt_cost, 2 * units * sizeof(uint32_t)) != hipSuccess) return 1;
                sb.lpt_cap = units;
            }
            sb.lpt_order = sb.lpt_cost + sb.lpt_cap;
            sb.lpt_units = units;
            sb.lpt_key = 0;
            if (hipMemsetAsync(sb.lpt_cost, 0, units * sizeof(uint32_t), stream) != hipSuccess) return 1;
        }
        uint64_t h = 1469598103934665603ull;
        auto mix = [&](const void *p, size_t n) { const unsigned char *b = (const unsigned char *)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } };
        (void)u;   // the view is deliberately not part of the key: a camera that moves a little keeps most of the cost map
        const uint32_t geo[8] = {Li.width, Li.height, Li.band_rows, Li.band_parts, Li.band_part, Li.n_local_rows, Li.compact_rows, (uint32_t)(uintptr_t)stream};
        mix(geo, sizeof geo);
        if (h == 0) h = 1;
        // the cost map changes slowly: the order is rebuilt after the first launch of a geometry and then every eighth
        // launch (the costs in between accumulate as maxima), which keeps the sort off most frames' critical path
        const bool known = sb.lpt_key == h;
        if (!known) sb.lpt_age = 0;
        Li.cost = sb.lpt_cost;
        Li.order = known ? sb.lpt_order : nullptr;
        Li.order_next = (sb.lpt_age % 8u == 0u) ? sb.lpt_order : nullptr;
        sb.lpt_age++;
        sb.lpt_key = h;
        return 0;
    };
    auto launch_frames = [&](const RtLaunch &L0, const char **kerr) -> int {
        auto frame_launch = [&](size_t i, uint32_t parts, uint32_t part) {
            RtLaunch Li = L0;
            Li.u = frames[i].u; Li.accum_index = frames[i].accum_index; Li.blend = frames[i].blend ? 1u : 0u;
            uint32_t rect[4];
            const bool any = octree_screen_rect(s->d, frames[i].u, f->width, f->height, rect);
            hot_tiles(rect, any, f->band_rows, f->band_parts * parts, Li.hot);
            if (parts > 1) {
                rt_frame fj = *f;
                fj.band_parts = f->band_parts * parts; fj.band_part = part * f->band_parts + f->band_part;
                Li.band_parts = fj.band_parts; Li.band_part = fj.band_part; Li.n_local_rows = band_rows_of(&fj);
                Li.sub_parts = parts; Li.sub_part = part;
            }
            return Li;
        };
        if (!wavefront && !persistent && K == 1) {
            // A launch that covers only a small part of the frame (one of several ranks' bands, a small window) cannot
            // keep the machine busy and still lasts as long as its longest path, so consecutive frames of a sequence are
            // traced together: `batch` frames per launch (see RtBatch).
            const size_t px = (size_t)f->width * L0.n_local_rows;
            size_t batch = px >= RT_FUSED_MIN_PIXELS ? RT_BATCH_BLOCK / 64u : RT_MAX_BATCH;
            if (s->max_batch > 0) batch = std::min<size_t>(batch, (size_t)s->max_batch);
            for (size_t i = 0; i < n_frames;) {
                const size_t nb = std::min(batch, n_frames - i);
                RtLaunch Li = frame_launch(i, 1, 0);
                if (attach_order(Li, 0, frames[i].u, st) != 0) return 1;
                if (nb < 2) {
                    int rc = rt_launch_render(s->d, Li, nullptr, nullptr, st, kerr, s->timing ? &s->sub[0].probe : nullptr);
                    if (rc != 0) return rc;
                    i++;
                    continue;
                }
                if (rt_batch_groups((uint32_t)nb) > 1u && s->stage_elems < nb * px) {
                    if (s->stage) { s->retired.push_back(s->stage); s->stage = nullptr; s->stage_elems = 0; }
                    if (hipMalloc(&s->stage, batch * px * 16) != hipSuccess) return 1;
                    s->stage_elems = batch * px;
                }
                RtBatch B{};
                for (size_t k = 0; k < nb; k++) { B.u[k] = frames[i + k].u; B.accum_index[k] = frames[i + k].accum_index; B.blend[k] = frames[i + k].blend ? 1u : 0u; }
                B.n = (uint32_t)nb; B.stage_stride = (uint32_t)px; B.stage = (float4 *)s->stage;
                int rc = rt_launch_render(s->d, Li, nullptr, nullptr, st, kerr, s->timing ? &s->sub[0].probe : nullptr, &B);
                if (rc != 0) return rc;
                i += nb;
            }
            return 0;
        }
        if (F > 1) {
            const size_t spx = (size_t)f->width * L0.n_local_rows;
            if (hipEventRecord(s->fork_ev, st) != hipSuccess) return 1;
            for (uint32_t c = 0; c < K * F; c++) if (hipStreamWaitEvent(s->sub[c].stream, s->fork_ev, 0) != hipSuccess) return 1;
            auto same_view = [](rt_uniforms a, rt_uniforms b) { a.time = 0.0f; b.time = 0.0f; return memcmp(&a, &b, sizeof a) == 0; };
            size_t i = 0;
            for (uint32_t group = 0; i < n_frames; group++) {
                const uint32_t phase = group % F, before = (phase + F - 1u) % F;
                RtStack G{};
                G.stride = (uint32_t)spx;
                while (G.n < V && i + G.n < n_frames && (G.n == 0 || same_view(frames[i].u, frames[i + G.n].u))) {
                    G.time[G.n] = frames[i + G.n].u.time; G.accum_index[G.n] = frames[i + G.n].accum_index; G.blend[G.n] = frames[i + G.n].blend ? 1u : 0u;
                    G.n++;
                }
                const float4 *planes = (const float4 *)s->stage + (size_t)phase * V * spx;
                for (uint32_t j = 0; j < K; j++) {
                    RtLaunch Li = frame_launch(i, K, j);
                    if (Li.n_local_rows == 0) continue;
                    rt_scene::Sub &sb = s->sub[phase * K + j];
                    RtLaunch Lb = Li;                                   // the blend into the target
                    Lb.color = L0.color;
                    Li.color = (float *)planes; Li.compact_rows = 1; Li.blend = 0; Li.accum_index = 1;
                    // the other planes hold what the last frame of the sequence leaves in them
                    if (i + G.n < n_frames) { Li.depth = nullptr; Li.hit_kind = nullptr; Li.hit_slot = nullptr; Li.hit_t = nullptr; Li.n_rays = nullptr; }
                    int rc = rt_launch_render(s->d, Li, wavefront ? &sb.ws : nullptr, persistent ? sb.pool : nullptr, sb.stream, kerr, s->timing ? &sb.probe : nullptr, nullptr,
                                              wavefront ? &G : nullptr);
                    if (rc != 0) return rc;
                    if (group > 0 && hipStreamWaitEvent(sb.stream, s->sub[before * K + j].done, 0) != hipSuccess) return 1;
                    if (rt_launch_blend(Lb, planes, G, sb.stream) != 0) return 1;
                    if (hipEventRecord(sb.done, sb.stream) != hipSuccess) return 1;
                }
                i += G.n;
            }
            for (uint32_t c = 0; c < K * F; c++) {
                // (a chain without rows has recorded nothing; waiting for a never-recorded event returns at once)
                if (hipStreamWaitEvent(st, s->sub[c].done, 0) != hipSuccess) return 1;
            }
            return 0;
        }
        if (K == 1) {
            for (size_t i = 0; i < n_frames; i++) {
                const bool wf = wavefront && frames[i].u.max_recursion_limit >= 1;
                const bool ps = persistent && frames[i].u.max_recursion_limit >= 1;
                RtLaunch Li = frame_launch(i, 1, 0);
                if (!wf && !ps && attach_order(Li, 0, frames[i].u, st) != 0) return 1;
                int rc = rt_launch_render(s->d, Li, wf ? &s->sub[0].ws : nullptr, ps ? s->sub[0].pool : nullptr, st, kerr, s->timing ? &s->sub[0].probe : nullptr);
                if (rc != 0) return rc;
            }
            return 0;
        }
        if (hipEventRecord(s->fork_ev, st) != hipSuccess) return 1;
        for (uint32_t j = 0; j < K; j++) {
            if (frame_launch(0, K, j).n_local_rows == 0) continue;
            if (hipStreamWaitEvent(s->sub[j].stream, s->fork_ev, 0) != hipSuccess) return 1;
            for (size_t i = 0; i < n_frames; i++) {
                const bool wf = wavefront && frames[i].u.max_recursion_limit >= 1;
                const bool ps = persistent && frames[i].u.max_recursion_limit >= 1;
                RtLaunch Li = frame_launch(i, K, j);
                if (!wf && !ps && attach_order(Li, j, frames[i].u, s->sub[j].stream) != 0) return 1;
                int rc = rt_launch_render(s->d, Li, wf ? &s->sub[j].ws : nullptr, ps ? s->sub[j].pool : nullptr, s->sub[j].stream, kerr, s->timing ? &s->sub[j].probe : nullptr);
                if (rc != 0) return rc;
            }
            if (hipEventRecord(s->sub[j].done, s->sub[j].stream) != hipSuccess) return 1;
            if (hipStreamWaitEvent(st, s->sub[j].done, 0) != hipSuccess) return 1;
        }
        return 0;
    };
    auto launch_all = [&](const RtLaunch &L0, const char **kerr) -> int {
        if (!measure) return launch_frames(L0, kerr);
        if (hipEventRecord(s->tuner.a, st) != hipSuccess) return 1;
        const int rc = launch_frames(L0, kerr);
        if (rc != 0 || hipEventRecord(s->tuner.b, st) != hipSuccess) return rc != 0 ? rc : 1;
        s->tuner.pending = cand; s->tuner.pending_tab = s->tuner.cur; s->tuner.pending_frames = (uint32_t)n_frames;
        return 0;
    };
    const size_t px = (size_t)f->width * f->height;                                            // ndc planes
    const size_t tpx = f->compact_rows ? (size_t)f->width * L.n_local_rows : px;              // target planes
    const char *kerr = nullptr;
    if (t->memory == RT_MEM_DEVICE) {
        L.ndc_x = f->ndc_x; L.ndc_y = f->ndc_y;
        L.color = t->color; L.depth = t->depth; L.hit_kind = t->hit_kind; L.hit_slot = t->hit_slot; L.hit_t = t->hit_t; L.n_rays = t->n_rays;
        if (launch_all(L, &kerr) != 0) return fail(RT_ENODEVICE, "kernel launch failed: %s", kerr ? kerr : "?");
        note_mode(wavefront ? RT_PIPELINE_WAVEFRONT : persistent ? RT_PIPELINE_PERSISTENT : RT_PIPELINE_MEGAKERNEL, L.near_first);
        if (int mrc = mark_use()) return mrc;
        if (t->sync) HIP_TRY(hipStreamSynchronize(st));
        return RT_OK;
    }
    // Host target: stage through device buffers owned by the scene.
    // (the two kinds of plane have capacities of their own: a compact band buffer and a full frame can hold the same
    // number of pixels in total and still need planes of different sizes)
    if (s->s_target_cap < tpx || s->s_ndc_cap < px || !s->s_color) {
        const size_t ncap = std::max(px, s->s_ndc_cap), tcap = std::max<size_t>(std::max(tpx, s->s_target_cap), 1);
        HIP_TRY(hipDeviceSynchronize());
        free_staging(s);
        HIP_TRY(hipMalloc(&s->s_color, tcap * 16)); HIP_TRY(hipMalloc(&s->s_depth, tcap * 4)); HIP_TRY(hipMalloc(&s->s_kind, tcap * 4));
        HIP_TRY(hipMalloc(&s->s_slot, tcap * 4)); HIP_TRY(hipMalloc(&s->s_t, tcap * 4)); HIP_TRY(hipMalloc(&s->s_nrays, tcap * 4));
        HIP_TRY(hipMalloc(&s->s_ndcx, ncap * 4)); HIP_TRY(hipMalloc(&s->s_ndcy, ncap * 4));
        s->s_ndc_cap = ncap; s->s_target_cap = tcap;
    }
    HIP_TRY(hipMemcpyAsync(s->s_color, t->color, tpx * 16, hipMemcpyHostToDevice, st));   // read-modify-write target (blend) and untouched bands
    if (f->ndc_x) {
        HIP_TRY(hipMemcpyAsync(s->s_ndcx, f->ndc_x, px * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(s->s_ndcy, f->ndc_y, px * 4, hipMemcpyHostToDevice, st));
        L.ndc_x = (const float *)s->s_ndcx; L.ndc_y = (const float *)s->s_ndcy;
    }
    struct Plane { void *host; void *dev; size_t bytes; };
    Plane planes[] = {{t->depth, s->s_depth, tpx * 4}, {t->hit_kind, s->s_kind, tpx * 4}, {t->hit_slot, s->s_slot, tpx * 4}, {t->hit_t, s->s_t, tpx * 4}, {t->n_rays, s->s_nrays, tpx * 4}};
    for (Plane &p : planes) if (p.host) HIP_TRY(hipMemcpyAsync(p.dev, p.host, p.bytes, hipMemcpyHostToDevice, st));
    L.color = (float *)s->s_color;
    L.depth = t->depth ? (float *)s->s_depth : nullptr; L.hit_kind = t->hit_kind ? (uint32_t *)s->s_kind : nullptr;
    L.hit_slot = t->hit_slot ? (uint32_t *)s->s_slot : nullptr; L.hit_t = t->hit_t ? (float *)s->s_t : nullptr; L.n_rays = t->n_rays ? (uint32_t *)s->s_nrays : nullptr;
    if (launch_all(L, &kerr) != 0) return fail(RT_ENODEVICE, "kernel launch failed: %s", kerr ? kerr : "?");
    HIP_TRY(hipMemcpyAsync(t->color, s->s_color, tpx * 16, hipMemcpyDeviceToHost, st));
    for (Plane &p : planes) if (p.host) HIP_TRY(hipMemcpyAsync(p.host, p.dev, p.bytes, hipMemcpyDeviceToHost, st));
    note_mode(wavefront ? RT_PIPELINE_WAVEFRONT : persistent ? RT_PIPELINE_PERSISTENT : RT_PIPELINE_MEGAKERNEL, L.near_first);
    if (int mrc = mark_use()) return mrc;
    HIP_TRY(hipStreamSynchronize(st));
    return RT_OK;
}

int rt_render(rt_scene *s, const rt_frame *f, const rt_target *t) { return render_frames(s, f, 1, t); }

int rt_render_sequence(rt_scene *s, const rt_frame *frames, size_t n_frames, const rt_target *t) { return render_frames(s, frames, n_frames, t); }

// A variant of rt_render for callers that cannot allocate (see rt_hotpath.h): always the one-kernel pipeline, its cost map
// and start order in this device's copy of `scratch`.  Nothing here allocates or frees: no hipMalloc / hipFree, no event or
// stream creation, no growing host container.
int rt_render_noalloc(rt_scene *s, const rt_frame *f, const rt_target *t) {
    if (!s || !f || !t) return fail(RT_EINVAL, "null argument");
    if (!t->color) return fail(RT_EINVAL, "target colour plane is null");
    if (t->memory != RT_MEM_DEVICE) return fail(RT_EINVAL, "rt_render_noalloc takes device-memory targets only (a host target needs staging buffers)");
    if (f->width == 0 || f->height == 0 || f->width > 65536 || f->height > 65536) return fail(RT_EINVAL, "frame size %ux%u out of range", f->width, f->height);
    if (f->band_parts == 0 || f->band_part >= f->band_parts || f->band_rows == 0) return fail(RT_EINVAL, "band selection %u/%u (rows %u) invalid", f->band_part, f->band_parts, f->band_rows);
    if ((f->ndc_x == nullptr) != (f->ndc_y == nullptr)) return fail(RT_EINVAL, "ndc_x and ndc_y must be given together");
    if ((unsigned long long)f->width * f->height >= 0xFFFFFFFFull) return fail(RT_ELIMIT, "frame has too many pixels");
    if (f->accum_index == 0) return fail(RT_EINVAL, "accum_index must be >= 1");
    if (f->compact_rows > 1) return fail(RT_EINVAL, "compact_rows must be 0 or 1");
    const uint32_t n_local_rows = band_rows_of(f);
    const size_t units = rt_render_units(f->width, n_local_rows);
    if (units > RT_NOALLOC_MAX_TILES)
        return fail(RT_ELIMIT, "rt_render_noalloc: this call covers %zu tiles of 16x16 pixels, the static scratch holds %u (use rt_render, or render the frame in bands)", units, RT_NOALLOC_MAX_TILES);
    DeviceGuard guard(s->device);
    if (!guard.ok) return fail(RT_ENODEVICE, "cannot select HIP device %d", s->device);
    if (s->device >= kMaxDevices || !g_noalloc[s->device].ptr) { if (int nrc = noalloc_setup(s->device)) return nrc; }
    NoAllocState &na = g_noalloc[s->device];
    hipStream_t st = (hipStream_t)f->stream;
    // Under stream capture the call becomes a graph node that is replayed later, any number of times, whatever other calls have
    // done to the shared array in between: a captured launch therefore neither reads nor writes the array (tiles start in index
    // order), and the scene's last-use event -- which orders work across streams outside a capture -- is left alone.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    const bool capturing = cap != hipStreamCaptureStatusNone;
    if (!capturing && s->used && s->last_stream != st) HIP_TRY(hipStreamWaitEvent(st, s->last_use, 0));

    RtLaunch L{};
    L.u = f->u;
    L.width = f->width; L.height = f->height; L.accum_index = f->accum_index; L.blend = f->blend ? 1u : 0u;
    L.band_rows = f->band_rows; L.band_parts = f->band_parts; L.band_part = f->band_part;
    L.n_local_rows = n_local_rows;
    L.compact_rows = f->compact_rows;
    L.near_first = s->order == RT_ORDER_NEAR_FIRST ? 1u : 0u;   // (the one-kernel pipeline has the reference's order and the plain near-first order)
    L.n_cu = (uint32_t)s->n_cu;
    L.sub_parts = 1; L.sub_part = 0;
    L.counters = s->counting ? s->d_counters : nullptr;
    if (s->counting) HIP_TRY(hipMemsetAsync(s->d_counters, 0, 128, st));
    L.ndc_x = f->ndc_x; L.ndc_y = f->ndc_y;
    L.color = t->color; L.depth = t->depth; L.hit_kind = t->hit_kind; L.hit_slot = t->hit_slot; L.hit_t = t->hit_t; L.n_rays = t->n_rays;
    uint32_t rect[4];
    const bool any = octree_screen_rect(s->d, f->u, f->width, f->height, rect);
    hot_tiles(rect, any, f->band_rows, f->band_parts, L.hot);
    if (units != 0) {
        const char *kerr = nullptr;
        if (capturing) {
            if (rt_launch_render(s->d, L, nullptr, nullptr, st, &kerr, nullptr, nullptr) != 0) return fail(RT_ENODEVICE, "kernel launch failed: %s", kerr ? kerr : "?");
        } else {
            // The array keeps what the last call on this device left, whichever scene that was for.  An order is a permutation
            // of the tiles of ONE frame geometry: it is used only by a call with the same geometry, and a call with another one
            // starts from zeroed costs and the index order.  (The bookkeeping is per device and guarded; the array's CONTENTS are
            // only safe when the calls on a device do not overlap in time, as the header says.)
            const uint32_t geo[7] = {f->width, f->height, f->band_rows, f->band_parts, f->band_part, n_local_rows, f->compact_rows};
            uint64_t h = 1469598103934665603ull;
            for (size_t i = 0; i < sizeof geo; i++) { h ^= ((const unsigned char *)geo)[i]; h *= 1099511628211ull; }
            if (h == 0) h = 1;
            std::lock_guard<std::mutex> lock(g_noalloc_mutex);
            uint32_t *cost = na.ptr, *order = na.ptr + RT_NOALLOC_MAX_TILES;
            const bool known = na.key == h;
            if (!known) {
                na.age = 0; na.key = 0;
                HIP_TRY(hipMemsetAsync(cost, 0, units * sizeof(uint32_t), st));
            }
            L.cost = cost;
            L.order = known ? order : nullptr;
            L.order_next = (na.age % 8u == 0u) ? order : nullptr;
            na.age++;
            // (the key is set only once a launch that writes the order is enqueued: age 0 always does)
            if (rt_launch_render(s->d, L, nullptr, nullptr, st, &kerr, nullptr, nullptr) != 0) { na.key = 0; return fail(RT_ENODEVICE, "kernel launch failed: %s", kerr ? kerr : "?"); }
            na.key = h;
        }
    }
    s->last_pipeline = RT_PIPELINE_MEGAKERNEL; s->last_order = (int)L.near_first;
    if (capturing) return RT_OK;     // (sync = 1 cannot be honoured inside a capture: the work runs when the graph is launched)
    if (hipEventRecord(s->last_use, st) != hipSuccess) return fail(RT_ENODEVICE, "hipEventRecord failed");
    s->last_stream = st; s->used = true;
    if (t->sync) HIP_TRY(hipStreamSynchronize(st));
    return RT_OK;
}

int rt_scene_destroy(rt_scene *s) {
    if (!s) return RT_OK;
    DeviceGuard guard(s->device);
    void *p[] = {s->d_vertices, s->d_normals, s->d_refs, s->d_nodes, s->d_tris, s->d_shade, s->d_dup, s->d_risky, s->d_planes, s->d_sky, s->d_counters};
    for (void *q : p) if (q) (void)hipFree(q);
    free_staging(s);
    for (rt_scene::Sub &sb : s->sub) {
        if (sb.stream) { (void)hipStreamSynchronize(sb.stream); (void)hipStreamDestroy(sb.stream); }
        if (sb.done) (void)hipEventDestroy(sb.done);
        if (sb.block) (void)hipFree(sb.block);
        if (sb.pool) (void)hipFree(sb.pool);
        if (sb.lpt_cost) (void)hipFree(sb.lpt_cost);
        if (sb.probe.events) { for (uint32_t i = 0; i < 2 * sb.probe.capacity; i++) if (sb.probe.events[i]) (void)hipEventDestroy(sb.probe.events[i]); delete[] sb.probe.events; delete[] sb.probe.launches; }
    }
    if (s->fork_ev) (void)hipEventDestroy(s->fork_ev);
    if (s->last_use) (void)hipEventDestroy(s->last_use);
    if (s->stage) (void)hipFree(s->stage);
    for (void *q : s->retired) (void)hipFree(q);
    if (s->tuner.a) (void)hipEventDestroy(s->tuner.a);
    if (s->tuner.b) (void)hipEventDestroy(s->tuner.b);
    delete s;
    return RT_OK;
}

}  // extern "C"
