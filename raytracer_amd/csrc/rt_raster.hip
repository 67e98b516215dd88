// The raster pass that follows the ray-tracing draw in the reference's frame (App.cpp:142-166), as HIP kernels behind
// include/rt_overlay.h: indexed triangle meshes through Vert_PosNormTex.vert:19-24 and Frag_LightingSimple.frag:29-41, back
// faces culled (App.cpp:50-51), depth test GL_LEQUAL with depth writes against the 24-bit attachment (App.cpp:54,
// Framebuffer.cpp:10-12), blending off.  Every value is produced by the operations, in the order and with the fusing, that
// Mesa llvmpipe uses for this pass (stated next to each step), so coverage, the 24-bit depth and the colour come out bit for bit.
//
// Four launches per call, none of them ordered by atomics on shared counters:
//   vertex_kernel    one thread per (draw, vertex): clip position, window position, the eight outputs
//   setup_kernel     one thread per (draw, triangle): clipping, then per resulting triangle the snapped corners, the bounding
//                    box and the plane equations, written to the triangle's own slots (RT_FAN_MAX per input triangle)
//   coverage_kernel  64 lanes per slot walk its bounding box and merge (depth24, draw order) into one 64-bit key per pixel
//                    with atomicMin: the smallest depth wins, and among equal depths the fragment drawn last, which is what
//                    GL_LEQUAL with depth writes leaves behind after the draws in their order
//   resolve_kernel   one thread per pixel: the winning slot's planes give the varyings, the fragment stage gives the colour
// The key plane starts from the attachment's current values, so the test against what the ray tracer left is part of the merge.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <new>
#include <vector>

#include "../../include/rt_overlay.h"

extern "C" int rt_internal_set_error(int code, const char *msg);

namespace {

#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { char m_[256]; snprintf(m_, sizeof m_, "%s: %s", #call, hipGetErrorString(e_)); return rt_internal_set_error(e_ == hipErrorOutOfMemory ? RT_ENOMEM : RT_ENODEVICE, m_); } } while (0)

constexpr unsigned BLOCK = 256;
constexpr unsigned RT_FAN_MAX = 7;        // a triangle clipped by six planes has at most nine corners: seven triangles
constexpr unsigned CHUNKS = 8;            // waves that share the bounding box of an ordinary triangle
constexpr unsigned BIG_TILES = 64;        // a triangle whose bounding box has more 8 x 8 tiles than this is listed as big ...
constexpr unsigned BIG_SPLIT = 128;       // ... and shared by this many waves: one wave's tiles are a chain of dependent loads and atomics
constexpr uint32_t BIG_CAPACITY = 1u << 20;
constexpr uint32_t NO_FRAGMENT = 0xFFFFFFFFu;

struct OutVertex { float pos[4]; float win[4]; float attr[8]; };      // clip position; window x, y, z and 1/w; world position, normal, texcoord

struct RasterTri {
    int32_t fx[3], fy[3];              // corners relative to the pixel centres, in 1/256 pixel
    int32_t x_lo, x_hi, y_lo, y_hi;    // pixels to test (x_lo > x_hi: nothing to draw)
    float a0[10], dadx[10], dady[10];  // z, 1/w and the eight outputs times 1/w
    uint32_t big, pad;                 // big: listed in the big-triangle list (the ordinary walk skips it)
};

constexpr int MAX_LEVELS = 16;
struct MipTable { uint32_t n_levels; uint32_t w[MAX_LEVELS], h[MAX_LEVELS], offset[MAX_LEVELS]; };      // offsets in texels into one buffer

struct PassConstants {
    float view_proj[16];
    float light[3], la[3], ld[3];
    uint32_t n_draws, n_vert, n_tri, width, height;
    uint32_t band_rows, band_parts, band_part, compact, plane_rows;      // which rows of the frame this call draws, and how the planes hold them
    MipTable mip;
};

// Rows of the frame are dealt out in strips of band_rows to band_parts owners (rt_frame's rule); a compact plane holds only its owner's strips.
__device__ __forceinline__ bool row_in_band(const PassConstants &C, uint32_t y) { return (y / C.band_rows) % C.band_parts == C.band_part; }
__device__ __forceinline__ uint32_t plane_row_of(const PassConstants &C, uint32_t y) { return C.compact ? (y / C.band_rows / C.band_parts) * C.band_rows + y % C.band_rows : y; }
__device__ __forceinline__ uint32_t frame_row_of(const PassConstants &C, uint32_t row) { return C.compact ? ((row / C.band_rows) * C.band_parts + C.band_part) * C.band_rows + row % C.band_rows : row; }

// mat4 * vec4 as the vertex stage evaluates it: ((c0*x + c1*y) + c2*z) + c3*w, nothing fused
__device__ __forceinline__ void mat_vec(const float *M, float x, float y, float z, float w, float *out) {
#pragma unroll
    for (int r = 0; r < 4; r++) out[r] = ((M[0 + r] * x + M[4 + r] * y) + M[8 + r] * z) + M[12 + r] * w;
}

// Window position.  The vertex stage's own viewport transform is one fused multiply-add per coordinate; the clipper computes
// the window position of the corners it creates as ((clip * (1/w)) * scale) + translate, not fused.
__device__ __forceinline__ void to_window(OutVertex &o, float sx, float sy, bool fused) {
    const float oow = 1.0f / o.pos[3];
    if (fused) {
        o.win[0] = fmaf(o.pos[0] * oow, sx, sx); o.win[1] = fmaf(o.pos[1] * oow, sy, sy); o.win[2] = fmaf(o.pos[2] * oow, 0.5f, 0.5f);
    } else {
        o.win[0] = o.pos[0] * oow * sx + sx; o.win[1] = o.pos[1] * oow * sy + sy; o.win[2] = o.pos[2] * oow * 0.5f + 0.5f;
    }
    o.win[3] = oow;
}

// Vert_PosNormTex.vert:19-24 (gl_Position is evaluated as viewProj * (world * position))
__global__ __launch_bounds__(BLOCK) void vertex_kernel(PassConstants C, const float *__restrict__ vertices, const float *__restrict__ world, const float *__restrict__ world_it,
                                                       OutVertex *__restrict__ out) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= C.n_draws * C.n_vert) return;
    const uint32_t d = i / C.n_vert, v = i % C.n_vert;
    const float *in = vertices + 8 * (size_t)v;
    float Wm[16], Wi[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { Wm[k] = world[16 * d + k]; Wi[k] = world_it[16 * d + k]; }
    OutVertex o;
    float wp[4], wn[4];
    mat_vec(Wm, in[0], in[1], in[2], 1.0f, wp);
    mat_vec(C.view_proj, wp[0], wp[1], wp[2], wp[3], o.pos);
    mat_vec(Wi, in[3], in[4], in[5], 0.0f, wn);
    o.attr[0] = wp[0]; o.attr[1] = wp[1]; o.attr[2] = wp[2];
    o.attr[3] = wn[0]; o.attr[4] = wn[1]; o.attr[5] = wn[2];
    o.attr[6] = in[6]; o.attr[7] = in[7];
    to_window(o, (float)C.width / 2.0f, (float)C.height / 2.0f, true);
    out[i] = o;
}

__device__ __forceinline__ float plane_distance(const OutVertex &v, int plane) {      // >= 0 inside; the order the planes are applied in
    switch (plane) {
    case 0: return v.pos[3] - v.pos[0];
    case 1: return v.pos[3] + v.pos[0];
    case 2: return v.pos[3] - v.pos[1];
    case 3: return v.pos[3] + v.pos[1];
    case 4: return v.pos[3] + v.pos[2];
    default: return v.pos[3] - v.pos[2];
    }
}

// One triangle for the rasteriser.  Corners minus the half-pixel centre are snapped to 1/256 pixel; a clockwise or empty triangle
// is dropped; the planes of a counter-clockwise triangle (v0, v1, v2) are set up from P = v1, Q = v0, R = v2 in float window
// coordinates: ooa = 1 / (dx01*dy20 - dx20*dy01) with dx01 = P.x - Q.x, dx20 = R.x - P.x; dadx = da01*(dy20*ooa) - da20*(dy01*ooa),
// dady = da20*(dx01*ooa) - da01*(dx20*ooa), a0 = a(P) - (dadx*(P.x - 0.5) + dady*(P.y - 0.5)); nothing fused.
__device__ void setup_triangle(const PassConstants &C, const OutVertex &v0, const OutVertex &v1, const OutVertex &v2, RasterTri &T, uint32_t slot, uint32_t *big_list) {
    const OutVertex *v[3] = {&v0, &v1, &v2};
    int64_t fx[3], fy[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        // (outside the frustum only by rounding: the clamp keeps the products of the edge functions inside 64 bits)
        const float x = fminf(fmaxf((v[i]->win[0] - 0.5f) * 256.0f, -1.0e9f), 1.0e9f), y = fminf(fmaxf((v[i]->win[1] - 0.5f) * 256.0f, -1.0e9f), 1.0e9f);
        fx[i] = (int64_t)__builtin_rintf(x); fy[i] = (int64_t)__builtin_rintf(y);
        T.fx[i] = (int32_t)fx[i]; T.fy[i] = (int32_t)fy[i];
    }
    T.x_lo = 1; T.x_hi = 0; T.y_lo = 1; T.y_hi = 0; T.big = 0;
    const int64_t area = (fx[1] - fx[0]) * (fy[2] - fy[0]) - (fx[2] - fx[0]) * (fy[1] - fy[0]);
    if (!(area > 0)) return;
    const int64_t minx = min(fx[0], min(fx[1], fx[2])), maxx = max(fx[0], max(fx[1], fx[2])), miny = min(fy[0], min(fy[1], fy[2])), maxy = max(fy[0], max(fy[1], fy[2]));
    const int64_t x_lo = max((minx + 255) >> 8, (int64_t)0), x_hi = min(maxx >> 8, (int64_t)C.width - 1), y_lo = max((miny + 255) >> 8, (int64_t)0), y_hi = min(maxy >> 8, (int64_t)C.height - 1);
    if (x_lo > x_hi || y_lo > y_hi) return;
    const float *P = v1.win, *Q = v0.win, *R = v2.win;
    const float dx01 = P[0] - Q[0], dy01 = P[1] - Q[1], dx20 = R[0] - P[0], dy20 = R[1] - P[1];
    const float ooa = 1.0f / (dx01 * dy20 - dx20 * dy01);
    const float dy20_ooa = dy20 * ooa, dy01_ooa = dy01 * ooa, dx01_ooa = dx01 * ooa, dx20_ooa = dx20 * ooa;
    const float xc = P[0] - 0.5f, yc = P[1] - 0.5f;
#pragma unroll
    for (int k = 0; k < 10; k++) {
        float aP, aQ, aR;
        if (k < 2) { aP = P[2 + k]; aQ = Q[2 + k]; aR = R[2 + k]; }
        else { aP = v1.attr[k - 2] * P[3]; aQ = v0.attr[k - 2] * Q[3]; aR = v2.attr[k - 2] * R[3]; }
        const float da01 = aP - aQ, da20 = aR - aP;
        T.dadx[k] = da01 * dy20_ooa - da20 * dy01_ooa;
        T.dady[k] = da20 * dx01_ooa - da01 * dx20_ooa;
        T.a0[k] = aP - (T.dadx[k] * xc + T.dady[k] * yc);
    }
    T.x_lo = (int32_t)x_lo; T.x_hi = (int32_t)x_hi; T.y_lo = (int32_t)y_lo; T.y_hi = (int32_t)y_hi;
    if ((uint64_t)((x_hi - x_lo + 8) >> 3) * (uint64_t)((y_hi - y_lo + 8) >> 3) > BIG_TILES) {
        const uint32_t at = atomicAdd(big_list, 1u);                 // big_list[0]: entries so far
        if (at < BIG_CAPACITY) { big_list[1 + at] = slot; T.big = 1; }      // (a full list leaves the triangle to the ordinary walk)
    }
}

// Clipping as the reference's GL does it: a triangle with a corner outside any of -w <= x, y, z <= w is clipped plane by plane
// (x <= w, x >= -w, y <= w, y >= -w, z >= -w, z <= w; only the planes some corner violates), walking the polygon from its first
// corner.  A new corner on an edge is a + t * (b - a) for the clip position and every output, a being the end of the edge nearer
// to the plane, t = d_a / (d_a - d_b), not fused.  The polygon p0 .. pn-1 is drawn as the triangles (p[i-1], p[i], p0); these pieces
// (not unclipped triangles) pass a cull test in float window coordinates before the snapped-coordinate one.
__global__ __launch_bounds__(64) void setup_kernel(PassConstants C, const uint32_t *__restrict__ indices, const OutVertex *__restrict__ vs, RasterTri *__restrict__ slots, uint32_t *__restrict__ big_list) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= C.n_draws * C.n_tri) return;
    const uint32_t d = i / C.n_tri, t = i % C.n_tri;
    const OutVertex *base = vs + (size_t)d * C.n_vert;
    RasterTri *out = slots + (size_t)i * RT_FAN_MAX;
    for (unsigned k = 0; k < RT_FAN_MAX; k++) { out[k].x_lo = 1; out[k].x_hi = 0; out[k].y_lo = 1; out[k].y_hi = 0; out[k].big = 0; }
    OutVertex a = base[indices[3 * t]], b = base[indices[3 * t + 1]], c = base[indices[3 * t + 2]];
    unsigned any_out = 0, all_out = 63;
    {
        const OutVertex *v[3] = {&a, &b, &c};
        for (int q = 0; q < 3; q++) {
            unsigned mask = 0;
            for (int pl = 0; pl < 6; pl++) if (!(plane_distance(*v[q], pl) >= 0.0f)) mask |= 1u << pl;
            any_out |= mask; all_out &= mask;
        }
    }
    if (!any_out) { setup_triangle(C, a, b, c, out[0], i * RT_FAN_MAX, big_list); return; }
    if (all_out) return;
    OutVertex p0[9], p1[9];
    OutVertex *src = p0, *dst = p1;
    p0[0] = a; p0[1] = b; p0[2] = c;
    int n = 3;
    const float sx = (float)C.width / 2.0f, sy = (float)C.height / 2.0f;
    for (int pl = 0; pl < 6 && n >= 3; pl++) {
        if (!(any_out & (1u << pl))) continue;
        int m = 0;
        const OutVertex *prev = &src[0];
        float dp_prev = plane_distance(*prev, pl);
        for (int k = 1; k <= n; k++) {
            const OutVertex *cur = &src[k % n];
            const float dp = plane_distance(*cur, pl);
            if (dp_prev >= 0.0f && m < 9) dst[m++] = *prev;
            if ((dp < 0.0f) != (dp_prev < 0.0f) && m < 9) {
                const bool from_prev = __builtin_fabsf(dp_prev) < __builtin_fabsf(dp);
                const OutVertex &A = from_prev ? *prev : *cur, &B = from_prev ? *cur : *prev;
                const float tt = from_prev ? dp_prev / (dp_prev - dp) : dp / (dp - dp_prev);
                OutVertex &o = dst[m++];
                for (int q = 0; q < 4; q++) o.pos[q] = A.pos[q] + tt * (B.pos[q] - A.pos[q]);
                for (int q = 0; q < 8; q++) o.attr[q] = A.attr[q] + tt * (B.attr[q] - A.attr[q]);
                to_window(o, sx, sy, false);
            }
            prev = cur; dp_prev = dp;
        }
        n = m;
        OutVertex *sw = src; src = dst; dst = sw;
    }
    for (int q = 2; q < n && q - 2 < (int)RT_FAN_MAX; q++) {
        // the pieces of a clipped triangle are culled first by the sign of (v0 - v2) x (v1 - v2) in FLOAT window coordinates (zero passes)
        const float *a = src[q - 1].win, *b = src[q].win, *c = src[0].win;
        const float ex = a[0] - c[0], ey = a[1] - c[1], fx = b[0] - c[0], fy = b[1] - c[1];
        if (ex * fy - ey * fx < 0.0f) continue;
        setup_triangle(C, src[q - 1], src[q], src[0], out[q - 2], i * RT_FAN_MAX + (uint32_t)(q - 2), big_list);
    }
}

__device__ __forceinline__ uint32_t to_depth24(float z) {      // round-to-nearest-even(float(clamp(z, 0, 1) * 16777215))
    return (uint32_t)__builtin_rintf(fminf(fmaxf(z, 0.0f), 1.0f) * 16777215.0f);
}

__global__ __launch_bounds__(BLOCK) void key_init_kernel(const uint32_t *__restrict__ depth24, unsigned long long *__restrict__ keys, size_t n) {
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) keys[i] = ((unsigned long long)depth24[i] << 32) | NO_FRAGMENT;
}

// A pixel centre belongs to the triangle when all three edge functions (integers, at the centre) are positive, or zero on a left
// or a bottom edge.  z = fma(dady, y, fma(dadx, x, a0)) at the integer pixel coordinates.  A wave takes the bounding box in tiles
// of 8 x 8 pixels, one lane per pixel: an edge function at a pixel is its value at the tile's corner (the same for the whole wave)
// plus the lane's own offset, and a tile that lies outside one of the edges altogether is skipped by the wave as a whole.
__device__ void cover_tiles(const PassConstants &C, const RasterTri &T, uint32_t s, uint32_t first, uint32_t step, unsigned long long *__restrict__ keys) {
    const int32_t lx = (int32_t)(threadIdx.x & 7u), ly = (int32_t)(threadIdx.x >> 3);
    const int32_t x_lo = T.x_lo, x_hi = T.x_hi, y_lo = T.y_lo, y_hi = T.y_hi;
    const uint32_t tiles_x = (uint32_t)(x_hi - x_lo + 8) >> 3, tiles = tiles_x * ((uint32_t)(y_hi - y_lo + 8) >> 3);
    if (first >= tiles) return;
    int64_t ex[3], ey[3], fx[3], fy[3], lane_part[3], lane_max[3]; bool incl[3];
#pragma unroll
    for (int i = 0; i < 3; i++) { fx[i] = T.fx[i]; fy[i] = T.fy[i]; }
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int j = (i + 1) % 3;
        ex[i] = fx[j] - fx[i]; ey[i] = fy[j] - fy[i];
        incl[i] = ey[i] < 0 || (ey[i] == 0 && ex[i] > 0);
        lane_part[i] = ex[i] * (ly * 256) - ey[i] * (lx * 256);
        lane_max[i] = (ex[i] > 0 ? ex[i] * (7 * 256) : 0) + (ey[i] < 0 ? -ey[i] * (7 * 256) : 0);
    }
    const float a0 = T.a0[0], dadx = T.dadx[0], dady = T.dady[0];
    const unsigned long long order = (unsigned long long)(NO_FRAGMENT - 1u - s);      // later slots win between equal depths
    for (uint32_t tile = first; tile < tiles; tile += step) {
        const int32_t x0 = x_lo + (int32_t)(tile % tiles_x) * 8, y0 = y_lo + (int32_t)(tile / tiles_x) * 8;
        int64_t base[3];
        bool tile_out = false;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            base[k] = ex[k] * ((int64_t)y0 * 256 - fy[k]) - ey[k] * ((int64_t)x0 * 256 - fx[k]);
            tile_out = tile_out || base[k] + lane_max[k] < 0;
        }
        if (tile_out) continue;
        const int32_t x = x0 + lx, y = y0 + ly;
        bool inside = x <= x_hi && y <= y_hi && row_in_band(C, (uint32_t)y);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int64_t e = base[k] + lane_part[k];
            inside = inside && (e > 0 || (e == 0 && incl[k]));
        }
        if (!inside) continue;
        const float z = fmaf(dady, (float)y, fmaf(dadx, (float)x, a0));
        const unsigned long long key = ((unsigned long long)to_depth24(z) << 32) | order;
        unsigned long long *p = keys + (size_t)plane_row_of(C, (uint32_t)y) * C.width + (size_t)x;
        if (key < *p) atomicMin(p, key);
    }
}

__global__ __launch_bounds__(64) void coverage_kernel(PassConstants C, const RasterTri *__restrict__ slots, unsigned long long *__restrict__ keys, uint32_t n_slots, const uint32_t *__restrict__ big_list) {
    for (uint32_t w = blockIdx.x; w < n_slots * CHUNKS; w += gridDim.x) {
        const uint32_t s = w / CHUNKS;
        const RasterTri &T = slots[s];
        if (T.x_lo > T.x_hi || T.big) continue;
        cover_tiles(C, T, s, w % CHUNKS, CHUNKS, keys);
    }
    const uint32_t n_big = min(big_list[0], BIG_CAPACITY);
    for (uint32_t w = blockIdx.x; w < n_big * BIG_SPLIT; w += gridDim.x) {
        const uint32_t s = big_list[1 + w / BIG_SPLIT];
        cover_tiles(C, slots[s], s, w % BIG_SPLIT, BIG_SPLIT, keys);
    }
}

__device__ __forceinline__ int lerp8(int a, int b, int w) { return a + (((b - a) * w + 128) >> 8); }
__device__ __forceinline__ int wrap(int i, int n) { const int r = i % n; return r < 0 ? r + n : r; }

// Texel coordinate in 1/256 texel, as the reference's GL computes it per axis: a power-of-two size scales the coordinate itself
// (GL_REPEAT then wraps the integer texel index); any other size takes the fraction of the coordinate first (kept below 1) and scales
// that.  Rounded to nearest even.  (A NaN ends at the constants; a coordinate beyond the integer range is clamped.)
__device__ __forceinline__ int texel_fixed(float c, int size) {
    if ((size & (size - 1)) == 0) return (int)__builtin_rintf(fminf(fmaxf(c * (float)size * 256.0f, -2.0e9f), 2.0e9f));
    const float f = fminf(c - floorf(c), 0.99999994f);
    return (int)__builtin_rintf(f * (float)size * 256.0f);
}

// One bilinear lookup with GL_REPEAT in the 8-bit fixed point of the reference's GL: the texel coordinate minus 128 gives texel (>> 8)
// and weight (& 255); lerp(a, b, w) = a + (((b - a) * w + 128) >> 8) along x, then along y.
__device__ void bilinear8(const uint8_t *__restrict__ tex, const MipTable &M, uint32_t level, float u, float v, int *out) {
    const int tw = (int)M.w[level], th = (int)M.h[level];
    const uchar4 *px = reinterpret_cast<const uchar4 *>(tex) + M.offset[level];
    const int su = texel_fixed(u, tw) - 128, sv = texel_fixed(v, th) - 128;
    const int i0 = wrap(su >> 8, tw), i1 = wrap((su >> 8) + 1, tw), j0 = wrap(sv >> 8, th), j1 = wrap((sv >> 8) + 1, th);
    const int wu = su & 255, wv = sv & 255;
    const uchar4 a = px[(size_t)j0 * tw + i0], b = px[(size_t)j0 * tw + i1], d = px[(size_t)j1 * tw + i0], e = px[(size_t)j1 * tw + i1];
    out[0] = lerp8(lerp8(a.x, b.x, wu), lerp8(d.x, e.x, wu), wv);
    out[1] = lerp8(lerp8(a.y, b.y, wu), lerp8(d.y, e.y, wu), wv);
    out[2] = lerp8(lerp8(a.z, b.z, wu), lerp8(d.z, e.z, wu), wv);
    out[3] = lerp8(lerp8(a.w, b.w, wu), lerp8(d.w, e.w, wu), wv);
}

// GL_LINEAR_MIPMAP_LINEAR / GL_LINEAR (GLUtils.cpp:217-219) as the reference's GL evaluates it.  uv0, uvx, uvy: the texture coordinates
// at the lower left pixel of the fragment's 2 x 2 quad, at the pixel to its right and at the pixel above it (shared by the four
// fragments of the quad).  rho^2 = max(|d(uv)/dx|^2, |d(uv)/dy|^2) in texels; lod = log2(rho^2) / 2 with the log2 read off the float's
// bits (exponent + mantissa fraction: linear between powers of two); lod <= 0 magnifies (level 0); else levels floor(lod) and the next,
// mixed by the 8-bit lerp with weight floor(fraction * 256); beyond the last level, the last level.  Result times 1/255.
__device__ void sample_texture(const uint8_t *__restrict__ tex, const MipTable &M, float u, float v, float2 uv0, float2 uvx, float2 uvy, float *out) {
    const float tw = (float)M.w[0], th = (float)M.h[0];
    const float a = (uvx.x - uv0.x) * tw, c = (uvx.y - uv0.y) * th, b = (uvy.x - uv0.x) * tw, d = (uvy.y - uv0.y) * th;
    const float rho2 = fmaxf(a * a + c * c, b * b + d * d);
    const float lod = (float)(__float_as_int(rho2) - 0x3f800000) * 0x1p-23f * 0.5f;
    int col[4];
    const int last = (int)M.n_levels - 1;
    if (!(lod > 0.0f)) bilinear8(tex, M, 0, u, v, col);
    else {
        const float fl = floorf(lod);
        if (fl >= (float)last) bilinear8(tex, M, (uint32_t)last, u, v, col);
        else {
            const int l0 = (int)fl, w8 = (int)floorf((lod - fl) * 256.0f);
            int c0[4], c1[4];
            bilinear8(tex, M, (uint32_t)l0, u, v, c0); bilinear8(tex, M, (uint32_t)l0 + 1u, u, v, c1);
#pragma unroll
            for (int k = 0; k < 4; k++) col[k] = lerp8(c0[k], c1[k], w8);
        }
    }
    const float k255 = 1.0f / 255.0f;
#pragma unroll
    for (int k = 0; k < 4; k++) out[k] = (float)col[k] * k255;
}

// Frag_LightingSimple.frag:29-41 with dot(a, b) = (a.z*b.z + a.y*b.y) + a.x*b.x, normalize(v) = v * (1 / sqrt(dot(v, v))),
// clamp = min(max(., 0), 1); a varying is fma(dady, y, fma(dadx, x, a0)) * (1 / (1/w interpolated the same way)).
__global__ __launch_bounds__(BLOCK) void resolve_kernel(PassConstants C, const RasterTri *__restrict__ slots, const unsigned long long *__restrict__ keys, const uint8_t *__restrict__ tex,
                                                        float4 *__restrict__ color, uint32_t *__restrict__ depth24) {
    const size_t n = (size_t)C.width * C.plane_rows, stride = (size_t)gridDim.x * BLOCK;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        const unsigned long long key = keys[i];
        const uint32_t low = (uint32_t)key;
        if (low == NO_FRAGMENT) continue;
        const RasterTri &T = slots[NO_FRAGMENT - 1u - low];
        const uint32_t px = (uint32_t)(i % C.width), py = frame_row_of(C, (uint32_t)(i / C.width));
        const float xf = (float)px, yf = (float)py;
        const float w = 1.0f / fmaf(T.dady[1], yf, fmaf(T.dadx[1], xf, T.a0[1]));
        float a[8];
#pragma unroll
        for (int k = 3; k < 8; k++) a[k] = fmaf(T.dady[2 + k], yf, fmaf(T.dadx[2 + k], xf, T.a0[2 + k])) * w;
        const float nl = 1.0f / sqrtf((a[5] * a[5] + a[4] * a[4]) + a[3] * a[3]);
        const float nx = a[3] * nl, ny = a[4] * nl, nz = a[5] * nl;
        const float mx = -C.light[0], my = -C.light[1], mz = -C.light[2];
        const float ml = 1.0f / sqrtf((mz * mz + my * my) + mx * mx);
        const float lx = mx * ml, ly = my * ml, lz = mz * ml;
        const float cosa = fminf(fmaxf((nz * lz + ny * ly) + nx * lx, 0.0f), 1.0f);
        // the texture coordinates at three pixels of this fragment's quad (covered or not: the planes extend beyond the triangle)
        float2 quv[3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const float qx = (float)((px & ~1u) + (q == 1 ? 1u : 0u)), qy = (float)((py & ~1u) + (q == 2 ? 1u : 0u));
            const float qw = 1.0f / fmaf(T.dady[1], qy, fmaf(T.dadx[1], qx, T.a0[1]));
            quv[q] = make_float2(fmaf(T.dady[8], qy, fmaf(T.dadx[8], qx, T.a0[8])) * qw, fmaf(T.dady[9], qy, fmaf(T.dadx[9], qx, T.a0[9])) * qw);
        }
        float tx[4];
        sample_texture(tex, C.mip, a[6], a[7], quv[0], quv[1], quv[2], tx);
        color[i] = make_float4((C.la[0] + cosa * C.ld[0]) * tx[0], (C.la[1] + cosa * C.ld[1]) * tx[1], (C.la[2] + cosa * C.ld[2]) * tx[2], 1.0f * tx[3]);
        depth24[i] = (uint32_t)(key >> 32);
    }
}

struct Buffer {
    void *p = nullptr; size_t bytes = 0;
    hipError_t reserve(size_t want) {
        if (want <= bytes) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; bytes = 0; if (e != hipSuccess) return e; }
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) bytes = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

}  // namespace

struct rt_overlay_mesh {
    int device = 0;
    uint32_t n_vert = 0, n_tri = 0;
    MipTable mip{};                                   // level 0 = the texture as given; the rest as glGenerateMipmap leaves them
    float *vertices = nullptr; uint32_t *indices = nullptr; uint8_t *texture = nullptr;      // texture: all levels, one after the other
    Buffer out_vertices, slots, keys, matrices, big_list, stage_color, stage_depth;      // grown on demand, kept between calls
};

namespace {
// Host side of GL_REPEAT-less lookups for the mip chain: the same fixed-point arithmetic as the device's, edges clamped.
int host_lerp8(int a, int b, int w) { return a + (((b - a) * w + 128) >> 8); }
int host_texel_fixed(float c, int size) {
    if ((size & (size - 1)) == 0) return (int)std::lrintf(c * (float)size * 256.0f);
    const float f = std::fmin(c - std::floor(c), 0.99999994f);
    return (int)std::lrintf(f * (float)size * 256.0f);
}
// glGenerateMipmap (GLUtils.cpp:215) as the reference's GL performs it: every texel of a level is ONE bilinear lookup of the level
// above at the texel's centre, edges clamped (for even sizes the mean of four texels with the rounding of the 8-bit lerp).
void build_mip_chain(const uint8_t *texture, uint32_t w, uint32_t h, MipTable &M, std::vector<uint8_t> &out) {
    out.assign(texture, texture + (size_t)w * h * 4);
    M.n_levels = 1; M.w[0] = w; M.h[0] = h; M.offset[0] = 0;
    while ((w > 1 || h > 1) && M.n_levels < (uint32_t)MAX_LEVELS) {
        const uint32_t sw = w, sh = h, soff = M.offset[M.n_levels - 1];
        w = w > 1 ? w / 2 : 1; h = h > 1 ? h / 2 : 1;
        const uint32_t doff = (uint32_t)(out.size() / 4);
        out.resize(out.size() + (size_t)w * h * 4);
        const uint8_t *src = out.data() + (size_t)soff * 4;
        uint8_t *dst = out.data() + (size_t)doff * 4;
        for (uint32_t j = 0; j < h; j++)
            for (uint32_t i = 0; i < w; i++) {
                const int su = host_texel_fixed((float)((i + 0.5) / w), (int)sw) - 128, sv = host_texel_fixed((float)((j + 0.5) / h), (int)sh) - 128;
                auto clampi = [](int a, int n) { return a < 0 ? 0 : a > n - 1 ? n - 1 : a; };
                const int i0 = clampi(su >> 8, (int)sw), i1 = clampi((su >> 8) + 1, (int)sw), j0 = clampi(sv >> 8, (int)sh), j1 = clampi((sv >> 8) + 1, (int)sh);
                const int wu = su & 255, wv = sv & 255;
                for (int c = 0; c < 4; c++) {
                    const int a = src[((size_t)j0 * sw + i0) * 4 + c], b = src[((size_t)j0 * sw + i1) * 4 + c], d = src[((size_t)j1 * sw + i0) * 4 + c], e = src[((size_t)j1 * sw + i1) * 4 + c];
                    dst[((size_t)j * w + i) * 4 + c] = (uint8_t)host_lerp8(host_lerp8(a, b, wu), host_lerp8(d, e, wu), wv);
                }
            }
        M.w[M.n_levels] = w; M.h[M.n_levels] = h; M.offset[M.n_levels] = doff;
        M.n_levels++;
    }
}

struct DeviceScope {
    int prev = -1; bool ok = false;
    explicit DeviceScope(int dev) { if (hipGetDevice(&prev) != hipSuccess) return; ok = prev == dev || hipSetDevice(dev) == hipSuccess; if (!ok) prev = -1; else if (prev == dev) prev = -1; }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};
}  // namespace

extern "C" int rt_overlay_mesh_create(const float *vertices, size_t n_vert, const uint32_t *indices, size_t n_idx, const uint8_t *texture_rgba, uint32_t tex_width, uint32_t tex_height,
                                      int device, rt_overlay_mesh **out) {
    if (!out) return rt_internal_set_error(RT_EINVAL, "rt_overlay_mesh_create: out is null");
    *out = nullptr;
    if (!vertices || !indices || !texture_rgba) return rt_internal_set_error(RT_EINVAL, "rt_overlay_mesh_create: null array");
    if (n_vert == 0 || n_idx == 0 || n_idx % 3 != 0) return rt_internal_set_error(RT_EINVAL, "rt_overlay_mesh_create: needs vertices and a multiple of three indices");
    if (n_vert > (1u << 24) || n_idx / 3 > (1u << 24)) return rt_internal_set_error(RT_ELIMIT, "rt_overlay_mesh_create: at most 2^24 vertices and 2^24 triangles");
    if (tex_width == 0 || tex_height == 0 || tex_width > 16384 || tex_height > 16384) return rt_internal_set_error(RT_EINVAL, "rt_overlay_mesh_create: texture size must be in [1, 16384]");
    for (size_t i = 0; i < n_idx; i++) if (indices[i] >= n_vert) return rt_internal_set_error(RT_ESCENE, "rt_overlay_mesh_create: index beyond the vertex array");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return rt_internal_set_error(RT_ENODEVICE, "no HIP device (there is no CPU fallback)");
    if (device < 0 || device >= count) return rt_internal_set_error(RT_ENODEVICE, "rt_overlay_mesh_create: no such HIP device");
    DeviceScope scope(device);
    if (!scope.ok) return rt_internal_set_error(RT_ENODEVICE, "rt_overlay_mesh_create: cannot select the HIP device");
    rt_overlay_mesh *m = new (std::nothrow) rt_overlay_mesh;
    if (!m) return rt_internal_set_error(RT_ENOMEM, "rt_overlay_mesh_create: out of host memory");
    m->device = device; m->n_vert = (uint32_t)n_vert; m->n_tri = (uint32_t)(n_idx / 3);
    std::vector<uint8_t> levels;
    build_mip_chain(texture_rgba, tex_width, tex_height, m->mip, levels);
    texture_rgba = levels.data();
    const size_t tex_bytes = levels.size();
    hipError_t e = hipMalloc(&m->vertices, n_vert * 32);
    if (e == hipSuccess) e = hipMalloc(&m->indices, n_idx * 4);
    if (e == hipSuccess) e = hipMalloc(&m->texture, tex_bytes);
    if (e == hipSuccess) e = hipMemcpy(m->vertices, vertices, n_vert * 32, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(m->indices, indices, n_idx * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(m->texture, texture_rgba, tex_bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        char msg[256]; snprintf(msg, sizeof msg, "rt_overlay_mesh_create: %s", hipGetErrorString(e));
        rt_overlay_mesh_destroy(m);
        return rt_internal_set_error(e == hipErrorOutOfMemory ? RT_ENOMEM : RT_ENODEVICE, msg);
    }
    *out = m;
    return RT_OK;
}

extern "C" int rt_overlay_mesh_destroy(rt_overlay_mesh *m) {
    if (!m) return RT_OK;
    DeviceScope scope(m->device);
    (void)hipDeviceSynchronize();
    if (m->vertices) (void)hipFree(m->vertices);
    if (m->indices) (void)hipFree(m->indices);
    if (m->texture) (void)hipFree(m->texture);
    m->out_vertices.release(); m->slots.release(); m->keys.release(); m->matrices.release(); m->big_list.release(); m->stage_color.release(); m->stage_depth.release();
    delete m;
    return RT_OK;
}

extern "C" int rt_overlay_draw(rt_overlay_mesh *m, const rt_overlay_pass *pass, float *color, uint32_t *depth24, uint32_t width, uint32_t height, int memory, void *stream) {
    return rt_overlay_draw_band(m, pass, color, depth24, width, height, 16, 1, 0, 0, memory, stream);
}

extern "C" int rt_overlay_draw_band(rt_overlay_mesh *m, const rt_overlay_pass *pass, float *color, uint32_t *depth24, uint32_t width, uint32_t height,
                                    uint32_t band_rows, uint32_t band_parts, uint32_t band_part, uint32_t compact_rows, int memory, void *stream) {
    if (!m || !pass) return rt_internal_set_error(RT_EINVAL, "rt_overlay_draw: mesh or pass is null");
    if (band_rows == 0 || band_parts == 0 || band_part >= band_parts) return rt_internal_set_error(RT_EINVAL, "rt_overlay_draw_band: band_rows and band_parts must be positive and band_part below band_parts");
    if (!color || !depth24) return rt_internal_set_error(RT_EINVAL, "rt_overlay_draw: null plane");
    if (memory != RT_MEM_HOST && memory != RT_MEM_DEVICE) return rt_internal_set_error(RT_EINVAL, "rt_overlay_draw: memory must be RT_MEM_HOST or RT_MEM_DEVICE");
    if (width == 0 || height == 0 || width > 32768 || height > 32768) return rt_internal_set_error(RT_EINVAL, "rt_overlay_draw: frame size must be in [1, 32768]");
    if (pass->n_draws == 0) return RT_OK;
    if (!pass->world || !pass->world_it) return rt_internal_set_error(RT_EINVAL, "rt_overlay_draw: null matrix array");
    const unsigned long long n_slots = (unsigned long long)pass->n_draws * m->n_tri * RT_FAN_MAX;
    if (n_slots >= (unsigned long long)NO_FRAGMENT - 1u) return rt_internal_set_error(RT_ELIMIT, "rt_overlay_draw: draws x triangles x 7 must stay below 2^32");
    if (memory == RT_MEM_DEVICE && ((uintptr_t)color & 15u)) return rt_internal_set_error(RT_EINVAL, "rt_overlay_draw: the colour plane must be 16-byte aligned");
    DeviceScope scope(m->device);
    if (!scope.ok) return rt_internal_set_error(RT_ENODEVICE, "rt_overlay_draw: cannot select the mesh's HIP device");
    hipStream_t st = static_cast<hipStream_t>(stream);
    uint32_t plane_rows = height;
    if (compact_rows) {      // the rows of this owner's strips (the last strip of the frame may be short)
        const uint32_t strips = (height + band_rows - 1) / band_rows;
        plane_rows = 0;
        for (uint32_t sidx = band_part; sidx < strips; sidx += band_parts) plane_rows += sidx + 1 == strips ? height - sidx * band_rows : band_rows;
        if (plane_rows == 0) return RT_OK;
    }
    const size_t n_px = (size_t)width * plane_rows;
    // scratch of an earlier call may still be in use by work enqueued on another stream
    const size_t want_v = (size_t)pass->n_draws * m->n_vert * sizeof(OutVertex), want_s = (size_t)n_slots * sizeof(RasterTri), want_k = n_px * 8, want_m = (size_t)pass->n_draws * 32 * 4;
    const size_t want_b = (size_t)(1u + (n_slots < BIG_CAPACITY ? (uint32_t)n_slots : BIG_CAPACITY)) * 4;
    if (want_v > m->out_vertices.bytes || want_s > m->slots.bytes || want_k > m->keys.bytes || want_m > m->matrices.bytes || want_b > m->big_list.bytes ||
        (memory == RT_MEM_HOST && (n_px * 16 > m->stage_color.bytes || n_px * 4 > m->stage_depth.bytes))) HIP_OK(hipDeviceSynchronize());
    HIP_OK(m->out_vertices.reserve(want_v)); HIP_OK(m->slots.reserve(want_s)); HIP_OK(m->keys.reserve(want_k)); HIP_OK(m->matrices.reserve(want_m)); HIP_OK(m->big_list.reserve(want_b));
    float4 *d_color = reinterpret_cast<float4 *>(color); uint32_t *d_depth = depth24;
    if (memory == RT_MEM_HOST) {
        HIP_OK(m->stage_color.reserve(n_px * 16)); HIP_OK(m->stage_depth.reserve(n_px * 4));
        d_color = static_cast<float4 *>(m->stage_color.p); d_depth = static_cast<uint32_t *>(m->stage_depth.p);
        HIP_OK(hipMemcpyAsync(d_color, color, n_px * 16, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(d_depth, depth24, n_px * 4, hipMemcpyHostToDevice, st));
    }
    float *d_world = static_cast<float *>(m->matrices.p), *d_world_it = d_world + (size_t)pass->n_draws * 16;
    HIP_OK(hipMemcpyAsync(d_world, pass->world, (size_t)pass->n_draws * 64, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(d_world_it, pass->world_it, (size_t)pass->n_draws * 64, hipMemcpyHostToDevice, st));
    HIP_OK(hipStreamSynchronize(st));      // the matrices are the caller's host memory: free to change once the call returns
    PassConstants C;
    for (int k = 0; k < 16; k++) C.view_proj[k] = pass->view_proj[k];
    for (int k = 0; k < 3; k++) { C.light[k] = pass->light_dir[k]; C.la[k] = pass->la[k]; C.ld[k] = pass->ld[k]; }
    C.n_draws = pass->n_draws; C.n_vert = m->n_vert; C.n_tri = m->n_tri; C.width = width; C.height = height; C.mip = m->mip;
    C.band_rows = band_rows; C.band_parts = band_parts; C.band_part = band_part; C.compact = compact_rows ? 1u : 0u; C.plane_rows = plane_rows;
    OutVertex *vs = static_cast<OutVertex *>(m->out_vertices.p);
    RasterTri *slots = static_cast<RasterTri *>(m->slots.p);
    unsigned long long *keys = static_cast<unsigned long long *>(m->keys.p);
    const unsigned px_grid = (unsigned)((n_px + BLOCK - 1) / BLOCK < 256u * 16u ? (n_px + BLOCK - 1) / BLOCK : 256u * 16u);
    if ((unsigned long long)pass->n_draws * m->n_vert >= (1ull << 32) || (unsigned long long)pass->n_draws * m->n_tri >= (1ull << 32))
        return rt_internal_set_error(RT_ELIMIT, "rt_overlay_draw: draws x vertices (and draws x triangles) must stay below 2^32");
    const uint32_t n_v = pass->n_draws * m->n_vert, n_t = pass->n_draws * m->n_tri;
    vertex_kernel<<<(n_v + BLOCK - 1) / BLOCK, BLOCK, 0, st>>>(C, m->vertices, d_world, d_world_it, vs);
    uint32_t *big_list = static_cast<uint32_t *>(m->big_list.p);
    HIP_OK(hipMemsetAsync(big_list, 0, 4, st));
    setup_kernel<<<(n_t + 63) / 64, 64, 0, st>>>(C, m->indices, vs, slots, big_list);
    key_init_kernel<<<px_grid, BLOCK, 0, st>>>(d_depth, keys, n_px);
    const unsigned long long work = n_slots * CHUNKS;
    coverage_kernel<<<(unsigned)(work < 256ull * 64ull ? work : 256ull * 64ull), 64, 0, st>>>(C, slots, keys, (uint32_t)n_slots, big_list);
    resolve_kernel<<<px_grid, BLOCK, 0, st>>>(C, slots, keys, m->texture, d_color, d_depth);
    HIP_OK(hipGetLastError());
    if (memory == RT_MEM_HOST) {
        HIP_OK(hipMemcpyAsync(color, d_color, n_px * 16, hipMemcpyDeviceToHost, st));
        HIP_OK(hipMemcpyAsync(depth24, d_depth, n_px * 4, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
    }
    return RT_OK;
}
