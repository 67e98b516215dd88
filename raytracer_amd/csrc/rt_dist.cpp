// The multi-GPU side of the C ABI (include/rt_dist.h): one process per GPU, scene broadcast and band gather over RCCL.
// No counterpart in the reference (one process, one GL context: app/main.cpp:19-174).  RCCL is loaded with dlopen at first use so
// that the library itself carries no link-time dependency on it.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>

#include "rt_internal.h"
#include "../../include/rt_dist.h"

static_assert(sizeof(ncclUniqueId) == RT_DIST_ID_BYTES, "id size");

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

int fail(int code, const char *fmt, const char *a = "", const char *b = "") {
    char buf[400];
    snprintf(buf, sizeof buf, fmt, a, b);
    return rt_internal_set_error(code, buf);
}

const Rccl *rccl() {
    std::call_once(g_rccl_once, [] {
        Rccl &r = g_rccl;
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) return;
        bool all = true;
        auto sym = [&](const char *n) { void *p = dlsym(r.lib, n); all = all && p != nullptr; return p; };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.Broadcast = (decltype(r.Broadcast))sym("ncclBroadcast");
        r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        r.ok = all;
    });
    return g_rccl.ok ? &g_rccl : nullptr;
}

struct DeviceGuard {
    int prev = -1; bool ok = false;
    explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess) ok = true; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
};

}  // namespace

struct rt_dist {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0, device = 0;
};

#define NCCL_TRY(expr)                                                                                    \
    do {                                                                                                  \
        ncclResult_t r_ = (expr);                                                                         \
        if (r_ != ncclSuccess) return fail(RT_ENODEVICE, "%s: %s", #expr, R->GetErrorString(r_));         \
    } while (0)
#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? RT_ENOMEM : RT_ENODEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

extern "C" {

int rt_dist_make_id(uint8_t id[RT_DIST_ID_BYTES]) {
    if (!id) return fail(RT_EINVAL, "id is null");
    const Rccl *R = rccl();
    if (!R) return fail(RT_ENODEVICE, "RCCL (librccl.so.1) cannot be loaded");
    ncclUniqueId u;
    NCCL_TRY(R->GetUniqueId(&u));
    memcpy(id, &u, RT_DIST_ID_BYTES);
    return RT_OK;
}

int rt_dist_init(const uint8_t id[RT_DIST_ID_BYTES], int world_size, int rank, int device, rt_dist **out) {
    if (!out) return fail(RT_EINVAL, "out is null");
    *out = nullptr;
    if (!id || world_size < 1 || rank < 0 || rank >= world_size) return fail(RT_EINVAL, "bad id, world size or rank");
    const Rccl *R = rccl();
    if (!R) return fail(RT_ENODEVICE, "RCCL (librccl.so.1) cannot be loaded");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return fail(RT_ENODEVICE, "no such HIP device");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(RT_ENODEVICE, "cannot select the HIP device");
    rt_dist *d = new (std::nothrow) rt_dist();
    if (!d) return fail(RT_ENOMEM, "out of host memory");
    d->world = world_size; d->rank = rank; d->device = device;
    ncclUniqueId u;
    memcpy(&u, id, RT_DIST_ID_BYTES);
    ncclResult_t r = R->CommInitRank(&d->comm, world_size, u, rank);
    if (r != ncclSuccess) { delete d; return fail(RT_ENODEVICE, "ncclCommInitRank: %s", R->GetErrorString(r)); }
    *out = d;
    return RT_OK;
}

int rt_dist_rank(const rt_dist *d) { return d ? d->rank : -1; }
int rt_dist_world_size(const rt_dist *d) { return d ? d->world : 0; }

int rt_dist_destroy(rt_dist *d) {
    if (!d) return RT_OK;
    const Rccl *R = rccl();
    if (R && d->comm) { DeviceGuard guard(d->device); (void)R->CommDestroy(d->comm); }
    delete d;
    return RT_OK;
}

namespace {
// The largest status code over all ranks (RT_OK = 0 is the smallest), so that every rank leaves a collective call with the same
// verdict.  `word`: four bytes of device memory.
int agree(const Rccl *R, rt_dist *d, void *word, int status, int *agreed) {
    HIP_TRY(hipMemcpy(word, &status, sizeof status, hipMemcpyHostToDevice));
    NCCL_TRY(R->AllReduce(word, word, 1, ncclInt, ncclMax, d->comm, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    HIP_TRY(hipMemcpy(agreed, word, sizeof status, hipMemcpyDeviceToHost));
    return RT_OK;
}
struct ThreadDevice {      // this thread's rt_set_device choice, for the duration of a call
    int prev;
    explicit ThreadDevice(int dev) : prev(rt_internal_swap_device(dev)) {}
    ~ThreadDevice() { (void)rt_internal_swap_device(prev); }
};

int scene_create_on_ranks(const Rccl *R, rt_dist *d, int root, const float *vertices, const float *normals, size_t n_vert, const uint32_t *tri_refs, size_t n_ref,
                          const uint32_t *nodes, size_t n_words, const float bounds_min[3], const float bounds_max[3], rt_scene **out) {
    const bool is_root = d->rank == root;
    // The root builds its scene first: that validates the arrays, and its verdict travels with the sizes so that every
    // rank returns the same code.
    struct Head { unsigned long long n_vert, n_ref, n_words; float lo[3], hi[3]; int status; int pad; } head{};
    int status = RT_OK;
    if (is_root) {
        if (!bounds_min || !bounds_max) status = fail(RT_EINVAL, "bounds are null");
        else status = rt_scene_create(vertices, normals, n_vert, tri_refs, n_ref, nodes, n_words, bounds_min, bounds_max, out);
        head.n_vert = n_vert; head.n_ref = n_ref; head.n_words = n_words; head.status = status;
        if (status == RT_OK) for (int k = 0; k < 3; k++) { head.lo[k] = bounds_min[k]; head.hi[k] = bounds_max[k]; }
    }
    DevBuf dh;
    HIP_TRY(dh.alloc(sizeof head));
    if (is_root) HIP_TRY(hipMemcpy(dh.p, &head, sizeof head, hipMemcpyHostToDevice));
    NCCL_TRY(R->Broadcast(dh.p, dh.p, sizeof head, ncclChar, root, d->comm, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    HIP_TRY(hipMemcpy(&head, dh.p, sizeof head, hipMemcpyDeviceToHost));
    if (head.status != RT_OK) {
        if (is_root) return status;                       // (its message is the root's own)
        return fail(head.status, "the root rank could not create the scene (its rt_last_error says why)");
    }
    // the four arrays, device to device: every rank first says whether it has the memory, so that none waits in a broadcast for
    // a rank that has given up
    const size_t bytes[4] = {(size_t)head.n_vert * 16, (size_t)head.n_vert * 16, (size_t)head.n_ref * 16, (size_t)head.n_words * 4};
    const void *src[4] = {vertices, normals, tri_refs, nodes};
    DevBuf buf[4];
    int local = RT_OK, all = RT_OK;
    for (int k = 0; k < 4 && local == RT_OK; k++) {
        hipError_t e = buf[k].alloc(bytes[k]);
        if (e == hipSuccess && is_root && bytes[k]) e = hipMemcpy(buf[k].p, src[k], bytes[k], hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipGetLastError(); local = fail(e == hipErrorOutOfMemory ? RT_ENOMEM : RT_ENODEVICE, "staging the scene arrays: %s", hipGetErrorString(e)); }
    }
    if (int rc = agree(R, d, dh.p, local, &all)) return rc;
    if (all != RT_OK) return local != RT_OK ? local : fail(all, "another rank could not stage the scene arrays (its rt_last_error says why)");
    for (int k = 0; k < 4; k++)
        if (bytes[k]) NCCL_TRY(R->Broadcast(buf[k].p, buf[k].p, bytes[k], ncclChar, root, d->comm, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    if (!is_root)
        local = rt_scene_create_from((const float *)buf[0].p, (const float *)buf[1].p, (size_t)head.n_vert, (const uint32_t *)buf[2].p, (size_t)head.n_ref,
                                     (const uint32_t *)buf[3].p, (size_t)head.n_words, head.lo, head.hi, 1, out);
    if (int rc = agree(R, d, dh.p, local, &all)) return rc;
    if (all != RT_OK) return local != RT_OK ? local : fail(all, "another rank could not derive its scene records (its rt_last_error says why)");
    return RT_OK;
}
}  // namespace

int rt_dist_scene_create(rt_dist *d, int root, const float *vertices, const float *normals, size_t n_vert, const uint32_t *tri_refs, size_t n_ref,
                         const uint32_t *nodes, size_t n_words, const float bounds_min[3], const float bounds_max[3], rt_scene **out) {
    if (!d || !out) return fail(RT_EINVAL, "null argument");
    *out = nullptr;
    if (root < 0 || root >= d->world) return fail(RT_EINVAL, "root rank out of range");
    const Rccl *R = rccl();
    if (!R) return fail(RT_ENODEVICE, "RCCL (librccl.so.1) cannot be loaded");
    DeviceGuard guard(d->device);
    if (!guard.ok) return fail(RT_ENODEVICE, "cannot select the HIP device");
    ThreadDevice chosen(d->device);      // the scene lands on this communicator's device whatever rt_set_device said before; restored on return
    const int rc = scene_create_on_ranks(R, d, root, vertices, normals, n_vert, tri_refs, n_ref, nodes, n_words, bounds_min, bounds_max, out);
    if (rc != RT_OK && *out) { (void)rt_scene_destroy(*out); *out = nullptr; }    // (rt_scene_destroy leaves the error message alone)
    return rc;
}

int rt_dist_skybox_set(rt_dist *d, int root, rt_scene *scene, const uint8_t *const faces[6], int w, int h) {
    if (!d || !scene) return fail(RT_EINVAL, "null argument");
    if (root < 0 || root >= d->world) return fail(RT_EINVAL, "root rank out of range");
    const Rccl *R = rccl();
    if (!R) return fail(RT_ENODEVICE, "RCCL (librccl.so.1) cannot be loaded");
    DeviceGuard guard(d->device);
    if (!guard.ok) return fail(RT_ENODEVICE, "cannot select the HIP device");
    const bool is_root = d->rank == root;
    int dims[4] = {w, h, 0, 0};
    if (is_root) {
        bool good = faces && w > 0 && h > 0 && w <= 16384 && h <= 16384;
        for (int f = 0; good && f < 6; f++) good = faces[f] != nullptr;
        dims[2] = good ? 1 : 0;
    }
    DevBuf dd;
    HIP_TRY(dd.alloc(sizeof dims));
    if (is_root) HIP_TRY(hipMemcpy(dd.p, dims, sizeof dims, hipMemcpyHostToDevice));
    NCCL_TRY(R->Broadcast(dd.p, dd.p, sizeof dims, ncclChar, root, d->comm, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    HIP_TRY(hipMemcpy(dims, dd.p, sizeof dims, hipMemcpyDeviceToHost));
    if (!dims[2]) return fail(RT_EINVAL, "skybox faces or size invalid on the root rank");
    const size_t fb = (size_t)dims[0] * dims[1] * 4;
    DevBuf all;
    HIP_TRY(all.alloc(fb * 6));
    if (is_root) for (int f = 0; f < 6; f++) HIP_TRY(hipMemcpy((char *)all.p + fb * f, faces[f], fb, hipMemcpyHostToDevice));
    NCCL_TRY(R->Broadcast(all.p, all.p, fb * 6, ncclChar, root, d->comm, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    const uint8_t *dev_faces[6];
    for (int f = 0; f < 6; f++) dev_faces[f] = (const uint8_t *)all.p + fb * f;
    return rt_skybox_set_from(scene, dev_faces, dims[0], dims[1], 1);
}

int rt_dist_frame_bands(const rt_dist *d, uint32_t band_rows, rt_frame *frame) {
    if (!d || !frame || band_rows == 0) return fail(RT_EINVAL, "bad argument");
    frame->band_rows = band_rows; frame->band_parts = (uint32_t)d->world; frame->band_part = (uint32_t)d->rank; frame->compact_rows = 1;
    return RT_OK;
}

int rt_dist_gather_frame(rt_dist *d, int root, const rt_frame *f, const void *band, void *frame_plane, size_t bpp, void *stream) {
    if (!d || !f || !band) return fail(RT_EINVAL, "null argument");
    if (root < 0 || root >= d->world || bpp == 0) return fail(RT_EINVAL, "root rank or pixel size invalid");
    if (f->band_rows == 0 || f->band_parts != (uint32_t)d->world || f->band_part != (uint32_t)d->rank || f->compact_rows != 1 || f->width == 0 || f->height == 0)
        return fail(RT_EINVAL, "the frame's band selection is not this communicator's (use rt_dist_frame_bands)");
    const Rccl *R = rccl();
    if (!R) return fail(RT_ENODEVICE, "RCCL (librccl.so.1) cannot be loaded");
    DeviceGuard guard(d->device);
    if (!guard.ok) return fail(RT_ENODEVICE, "cannot select the HIP device");
    hipStream_t st = (hipStream_t)stream;
    const bool is_root = d->rank == root;
    if (is_root && !frame_plane) return fail(RT_EINVAL, "the root rank needs the frame plane");
    const size_t row_bytes = (size_t)f->width * bpp, strip_bytes = row_bytes * f->band_rows;
    const uint32_t n_strips = (f->height + f->band_rows - 1) / f->band_rows, parts = (uint32_t)d->world;
    auto strip_rows = [&](uint32_t strip) { const uint32_t y0 = strip * f->band_rows; return y0 + f->band_rows <= f->height ? f->band_rows : f->height - y0; };
    // Strip s of the frame belongs to part s % parts and is that part's local strip s / parts.  Every strip travels as one
    // message straight to its rows of the root's frame; the root's own strips are one strided device copy.
    // (an error inside the group must not leave it open: the first one is kept and the group is closed whatever happened)
    NCCL_TRY(R->GroupStart());
    ncclResult_t first = ncclSuccess;
    for (uint32_t s = 0; s < n_strips && first == ncclSuccess; s++) {
        const uint32_t part = s % parts, local = s / parts;
        const size_t bytes = (size_t)strip_rows(s) * row_bytes;
        if (part == (uint32_t)root) continue;
        if (is_root) first = R->Recv((char *)frame_plane + (size_t)s * strip_bytes, bytes, ncclChar, (int)part, d->comm, st);
        else if (part == (uint32_t)d->rank) first = R->Send((const char *)band + (size_t)local * strip_bytes, bytes, ncclChar, root, d->comm, st);
    }
    const ncclResult_t closed = R->GroupEnd();
    if (first != ncclSuccess) return fail(RT_ENODEVICE, "ncclSend / ncclRecv of a strip: %s", R->GetErrorString(first));
    if (closed != ncclSuccess) return fail(RT_ENODEVICE, "ncclGroupEnd: %s", R->GetErrorString(closed));
    if (is_root) {
        for (uint32_t s = (uint32_t)root; s < n_strips; s += parts)
            HIP_TRY(hipMemcpyAsync((char *)frame_plane + (size_t)s * strip_bytes, (const char *)band + (size_t)(s / parts) * strip_bytes, (size_t)strip_rows(s) * row_bytes, hipMemcpyDeviceToDevice, st));
    }
    return RT_OK;
}

}  // extern "C"
