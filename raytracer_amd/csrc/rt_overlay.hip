// The depth-attachment side of the reference's overlay pass and the presentation conversion (include/rt_overlay.h): three streaming
// kernels, bound by HBM bandwidth (8 bytes per pixel for the depth conversion; 9 to 44 for the merge, depending on how many fragments
// pass; 20 for the 8-bit image).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "../../include/rt_overlay.h"

extern "C" int rt_internal_set_error(int code, const char *msg);

namespace {

#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { char m_[256]; snprintf(m_, sizeof m_, "%s: %s", #call, hipGetErrorString(e_)); return rt_internal_set_error(e_ == hipErrorOutOfMemory ? RT_ENOMEM : RT_ENODEVICE, m_); } } while (0)

constexpr unsigned BLOCK = 256;

// Framebuffer.cpp:10-12: float depth -> 24-bit unsigned normalised, as the depth attachment stores it.
__device__ __forceinline__ uint32_t to_depth24(float d) {
    const float c = fminf(fmaxf(d, 0.0f), 1.0f);                 // GL clamps the fragment depth; NaN becomes 0
    return (uint32_t)__builtin_rintf(c * 16777215.0f);           // the product rounded to float first, then ties to even
}

__global__ __launch_bounds__(BLOCK) void depth_to_attachment_kernel(const float *__restrict__ depth, uint32_t *__restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * BLOCK;
    const size_t n4 = n / 4;                                      // 16 bytes per lane per access
    if ((((uintptr_t)depth | (uintptr_t)out) & 15u) == 0) {
        for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n4; i += stride) {
            const float4 d = reinterpret_cast<const float4 *>(depth)[i];
            reinterpret_cast<uint4 *>(out)[i] = make_uint4(to_depth24(d.x), to_depth24(d.y), to_depth24(d.z), to_depth24(d.w));
        }
        for (size_t i = 4 * n4 + (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) out[i] = to_depth24(depth[i]);
    } else {
        for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) out[i] = to_depth24(depth[i]);
    }
}

// App.cpp:54,142-145: GL_LEQUAL depth test, depth writes, no blending.
__global__ __launch_bounds__(BLOCK) void overlay_composite_kernel(float4 *__restrict__ color, uint32_t *__restrict__ depth24, const float4 *__restrict__ frag_color,
                                                                  const uint32_t *__restrict__ frag_depth24, const uint8_t *__restrict__ frag_mask, size_t n) {
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        if (frag_mask && frag_mask[i] == 0) continue;
        const uint32_t z = frag_depth24[i];
        if (z <= depth24[i]) {
            color[i] = frag_color[i];
            depth24[i] = z;
        }
    }
}

// Framebuffer.cpp:33-35: float colour -> the window's 8-bit unsigned normalised components.
__device__ __forceinline__ uint32_t to_unorm8(float c) {
    return (uint32_t)__builtin_rintf(fminf(fmaxf(c, 0.0f), 1.0f) * 255.0f);      // NaN becomes 0; the product is a float before the conversion
}
__global__ __launch_bounds__(BLOCK) void present_kernel(const float4 *__restrict__ color, uint32_t *__restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        const float4 c = color[i];
        out[i] = to_unorm8(c.x) | (to_unorm8(c.y) << 8) | (to_unorm8(c.z) << 16) | (to_unorm8(c.w) << 24);
    }
}

unsigned grid_for(size_t work_items) {
    const size_t blocks = (work_items + BLOCK - 1) / BLOCK;
    return (unsigned)(blocks < 1 ? 1 : blocks > 256u * 16u ? 256u * 16u : blocks);   // up to 16 workgroups per CU, grid-stride beyond
}

// Device copies of host arrays for the RT_MEM_HOST form of the calls.
struct Staged {
    void *p[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int n = 0;
    ~Staged() { for (int i = 0; i < n; i++) (void)hipFree(p[i]); }
    hipError_t up(const void *src, size_t bytes, void **dst) {
        hipError_t e = hipMalloc(dst, bytes ? bytes : 1);
        if (e != hipSuccess) return e;
        p[n++] = *dst;
        return src && bytes ? hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) : hipSuccess;
    }
};

// Makes the device that owns `p` current for the lifetime of the object (planes may live on another GPU than the
// calling thread's current one); restores the previous device afterwards.
struct OnDeviceOf {
    int prev = -1, dev = -1;
    explicit OnDeviceOf(const void *p) {
        hipPointerAttribute_t at;
        if (hipGetDevice(&prev) != hipSuccess) { prev = -1; return; }
        if (hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeDevice) dev = at.device;
        else (void)hipGetLastError();
        if (dev >= 0 && dev != prev) (void)hipSetDevice(dev);
    }
    ~OnDeviceOf() { if (prev >= 0 && dev >= 0 && dev != prev) (void)hipSetDevice(prev); }
};

int device_ready() {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return rt_internal_set_error(RT_ENODEVICE, "no HIP device (there is no CPU fallback)");
    return RT_OK;
}

}  // namespace

extern "C" int rt_depth_to_attachment(const float *depth, uint32_t *depth24, size_t n, int memory, void *stream) {
    if ((!depth || !depth24) && n) return rt_internal_set_error(RT_EINVAL, "rt_depth_to_attachment: null plane");
    if (memory != RT_MEM_HOST && memory != RT_MEM_DEVICE) return rt_internal_set_error(RT_EINVAL, "rt_depth_to_attachment: memory must be RT_MEM_HOST or RT_MEM_DEVICE");
    if (const int rc = device_ready()) return rc;
    if (n == 0) return RT_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (memory == RT_MEM_DEVICE) {
        OnDeviceOf guard(depth24);
        depth_to_attachment_kernel<<<grid_for((n + 3) / 4), BLOCK, 0, st>>>(depth, depth24, n);
        HIP_OK(hipGetLastError());
        return RT_OK;
    }
    Staged s;
    void *d_in = nullptr, *d_out = nullptr;
    HIP_OK(s.up(depth, n * 4, &d_in));
    HIP_OK(s.up(nullptr, n * 4, &d_out));
    depth_to_attachment_kernel<<<grid_for((n + 3) / 4), BLOCK, 0, st>>>(static_cast<const float *>(d_in), static_cast<uint32_t *>(d_out), n);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipMemcpy(depth24, d_out, n * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

extern "C" int rt_overlay_composite(float *color, uint32_t *depth24, const float *frag_color, const uint32_t *frag_depth24, const uint8_t *frag_mask, size_t n, int memory, void *stream) {
    if ((!color || !depth24 || !frag_color || !frag_depth24) && n) return rt_internal_set_error(RT_EINVAL, "rt_overlay_composite: null plane");
    if (memory != RT_MEM_HOST && memory != RT_MEM_DEVICE) return rt_internal_set_error(RT_EINVAL, "rt_overlay_composite: memory must be RT_MEM_HOST or RT_MEM_DEVICE");
    if (const int rc = device_ready()) return rc;
    if (n == 0) return RT_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (memory == RT_MEM_DEVICE) {
        if (((uintptr_t)color | (uintptr_t)frag_color) & 15u) return rt_internal_set_error(RT_EINVAL, "rt_overlay_composite: colour planes must be 16-byte aligned");
        OnDeviceOf guard(depth24);
        overlay_composite_kernel<<<grid_for(n), BLOCK, 0, st>>>(reinterpret_cast<float4 *>(color), depth24, reinterpret_cast<const float4 *>(frag_color), frag_depth24, frag_mask, n);
        HIP_OK(hipGetLastError());
        return RT_OK;
    }
    Staged s;
    void *d_color = nullptr, *d_z = nullptr, *d_fc = nullptr, *d_fz = nullptr, *d_mask = nullptr;
    HIP_OK(s.up(color, n * 16, &d_color));
    HIP_OK(s.up(depth24, n * 4, &d_z));
    HIP_OK(s.up(frag_color, n * 16, &d_fc));
    HIP_OK(s.up(frag_depth24, n * 4, &d_fz));
    if (frag_mask) HIP_OK(s.up(frag_mask, n, &d_mask));
    overlay_composite_kernel<<<grid_for(n), BLOCK, 0, st>>>(static_cast<float4 *>(d_color), static_cast<uint32_t *>(d_z), static_cast<const float4 *>(d_fc), static_cast<const uint32_t *>(d_fz),
                                                            static_cast<const uint8_t *>(d_mask), n);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipMemcpy(color, d_color, n * 16, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(depth24, d_z, n * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

extern "C" int rt_present_rgba8(const float *color, uint8_t *rgba8, size_t n, int memory, void *stream) {
    if ((!color || !rgba8) && n) return rt_internal_set_error(RT_EINVAL, "rt_present_rgba8: null plane");
    if (memory != RT_MEM_HOST && memory != RT_MEM_DEVICE) return rt_internal_set_error(RT_EINVAL, "rt_present_rgba8: memory must be RT_MEM_HOST or RT_MEM_DEVICE");
    if (const int rc = device_ready()) return rc;
    if (n == 0) return RT_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (memory == RT_MEM_DEVICE) {
        if (((uintptr_t)color & 15u) || ((uintptr_t)rgba8 & 3u)) return rt_internal_set_error(RT_EINVAL, "rt_present_rgba8: the colour plane must be 16-byte aligned and the byte plane 4-byte aligned");
        OnDeviceOf guard(rgba8);
        present_kernel<<<grid_for(n), BLOCK, 0, st>>>(reinterpret_cast<const float4 *>(color), reinterpret_cast<uint32_t *>(rgba8), n);
        HIP_OK(hipGetLastError());
        return RT_OK;
    }
    Staged s;
    void *d_in = nullptr, *d_out = nullptr;
    HIP_OK(s.up(color, n * 16, &d_in));
    HIP_OK(s.up(nullptr, n * 4, &d_out));
    present_kernel<<<grid_for(n), BLOCK, 0, st>>>(static_cast<const float4 *>(d_in), static_cast<uint32_t *>(d_out), n);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipMemcpy(rgba8, d_out, n * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}
