// Device code of the ray-tracing pass for gfx950 (MI355X).
//
// What the reference's fragment shader app/assets/ray_tracer.frag does per pixel (cited as frag:N):
// ray generation frag:674-687, closest hit over 84 analytic spheres / the triangle octree / one cylinder /
// two portals frag:427-563, the path loop with Lambertian / metal / dielectric scattering frag:566-665,
// environment lookup frag:662, filmic tonemap frag:667-668 and the accumulate-blend of
// app/core/src/App.cpp:49,104.  Results must equal the reference's CPU execution (Mesa llvmpipe on x86):
// IEEE binary32, flush-to-zero, NO contraction except inside the sin/cos/log2/exp2 polynomials (written as
// explicit fmaf), dot products summed z,y,x.  Compile with -ffp-contract=off -fgpu-flush-denormals-to-zero.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rt_internal.h"

#pragma clang fp contract(off)

#define DEVI __device__ __forceinline__

namespace {

// ------------------------------------------------------------------------------------------------
// small vector helpers (no fusing, fixed evaluation order)
// ------------------------------------------------------------------------------------------------
struct f3 { float x, y, z; };
DEVI f3 F3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
DEVI f3 operator+(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
DEVI f3 operator-(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
DEVI f3 operator*(f3 a, f3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }
DEVI f3 operator*(f3 a, float s) { return F3(a.x * s, a.y * s, a.z * s); }
DEVI f3 operator-(f3 a) { return F3(-a.x, -a.y, -a.z); }
DEVI float dot3(f3 a, f3 b) { return (a.z * b.z + a.y * b.y) + a.x * b.x; }
DEVI f3 cross3(f3 a, f3 b) { return F3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
// IEEE-correct square root, divide and reciprocal: hipcc's default (-fhip-fp32-correctly-rounded-divide-sqrt)
DEVI float sqrt_slow(float x) { return __builtin_sqrtf(x); }
DEVI float rcp_slow(float x) { return 1.0f / x; }
DEVI float div_slow(float a, float b) { return a / b; }
#ifdef RT_SLOW_MATH
DEVI float sqrt_ieee(float x) { return sqrt_slow(x); }
DEVI float rcp_ieee(float x) { return rcp_slow(x); }
#else
// The compiler's correctly rounded 1/x and sqrt(x) are sequences of about a dozen instructions between two switches of
// the denormal mode.  The hardware's own approximations (v_rcp_f32, v_sqrt_f32, v_rsq_f32: 1 ulp) followed by ONE Newton
// step in fused arithmetic give the same bits for EVERY one of the 2^32 inputs under this file's arithmetic mode
// (denormal inputs and results flushed) -- checked exhaustively on the device by rt_internal_math_check, which the GPU
// tests run.  The residuals are exact (fused), so the step lands within a rounding error far smaller than the distance of
// 1/x or sqrt(x) from a rounding boundary; zero, infinity and NaN make the residual NaN and keep the approximation.
DEVI float rcp_ieee(float x) {
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r0, 1.0f);
    const float r1 = __builtin_fmaf(e, r0, r0);
    return e != e ? r0 : r1;
}
DEVI float sqrt_ieee(float x) {
    // below 2^-95 the residual of the step would be flushed: such inputs are scaled by 2^64 (exact) and the root by 2^-32
    const bool small = x < 0x1p-64f;
    const float xs = small ? x * 0x1p+64f : x;
    const float s0 = __builtin_amdgcn_sqrtf(xs);
    const float h = 0.5f * __builtin_amdgcn_rsqf(xs);
    const float e = __builtin_fmaf(-s0, s0, xs);
    const float s1 = __builtin_fmaf(e, h, s0);
    const float s = s1 != s1 ? s0 : s1;
    return small ? s * 0x1p-32f : s;
}
#endif
DEVI float div_ieee(float a, float b) { return div_slow(a, b); }
DEVI float length3(f3 a) { return sqrt_ieee(dot3(a, a)); }
DEVI f3 normalize3(f3 a) { float inv = rcp_ieee(sqrt_ieee(dot3(a, a))); return a * inv; }
DEVI f3 rcp3(f3 a) { return F3(rcp_ieee(a.x), rcp_ieee(a.y), rcp_ieee(a.z)); }
// Identity the optimiser cannot see through: a value derived from it is computed where it is used instead of being
// hoisted out of the path loop and kept (or spilled) across the whole octree traversal.
DEVI uint32_t opaque(uint32_t x) { asm volatile("" : "+v"(x)); return x; }
DEVI uint32_t f2u(float f) { return __float_as_uint(f); }
DEVI float u2f(uint32_t u) { return __uint_as_float(u); }
DEVI bool is_inf(float x) { return __builtin_fabsf(x) == __builtin_inff(); }

// ------------------------------------------------------------------------------------------------
// transcendental functions exactly as the reference's CPU vehicle evaluates them
// ------------------------------------------------------------------------------------------------
template <bool IS_SIN> DEVI float sincos_poly(float x) {
    const float xa = __builtin_fabsf(x);
    int j = (int)(xa * 1.27323954473516f);   // 4/pi
    j = (j + 1) & ~1;
    const float yf = (float)j;
    bool negate, sin_poly;
    if (IS_SIN) { negate = ((j & 4) != 0) != (x < 0.0f); sin_poly = (j & 2) == 0; }
    else { const int j3 = j - 2; negate = (~j3 & 4) != 0; sin_poly = (j3 & 2) == 0; }
    float r = fmaf(yf, -0.78515625f, xa);
    r = fmaf(yf, -2.4187564849853515625e-4f, r);
    r = fmaf(yf, -3.77489497744594108e-8f, r);
    const float z = r * r;
    float s = fmaf(z, -1.9515295891E-4f, 8.3321608736E-3f);
    s = fmaf(s, z, -1.6666654611E-1f);
    s = s * z;
    s = fmaf(s, r, r);
    float c = fmaf(z, 2.443315711809948E-005f, -1.388731625493765E-003f);
    c = fmaf(c, z, 4.166664568298827E-002f);
    c = c * z;
    c = c * z;
    c = c - z * 0.5f;
    c = c + 1.0f;
    const float out = sin_poly ? s : c;
    return negate ? -out : out;
}
DEVI float log2_poly(float v) {
    if (v == 0.0f) return -__builtin_inff();
    const int i = (int)f2u(v);
    const float ex = (float)(((i >> 23) & 0xff) - 127);
    const float m = u2f(((uint32_t)i & 0x007fffffu) | 0x3f800000u);
    const float y = div_ieee(m - 1.0f, m + 1.0f);
    const float z = y * y, z2 = z * z;
    const float even = fmaf(z2, fmaf(z2, (float)0.40671805249884624, (float)0.5774403394387364), (float)2.8853900934330916);
    const float odd = fmaf(z2, (float)0.4033438582513299, (float)0.9617915504041842);
    return fmaf(y, fmaf(odd, z, even), ex);
}
DEVI float exp2_poly(float v) {
    v = fminf(fmaxf(v, -126.99999f), 129.0f);
    const float ip = __builtin_floorf(v), fp = v - ip;
    const float scale = u2f((uint32_t)((int)ip + 127) << 23);   // (in unsigned arithmetic: 256 << 23 for an argument of 129 wraps to the sign bit, as pslld does)
    const float f2 = fp * fp;
    const float even = fmaf(f2, fmaf(f2, (float)0.008989340090494664, (float)0.24015361704437538), 1.0f);
    const float odd = fmaf(f2, fmaf(f2, (float)0.001877576675191479, (float)0.055826318053295666), (float)0.693153073200169);
    return scale * fmaf(odd, fp, even);
}
DEVI float pow_poly(float x, float y) { return exp2_poly(log2_poly(x) * y); }

// frag:359-363
DEVI uint32_t base_hash(uint32_t px, uint32_t py) {
    const uint32_t qx = 1103515245u * ((px >> 1) ^ py);
    const uint32_t qy = 1103515245u * ((py >> 1) ^ px);
    const uint32_t h = 1103515245u * (qx ^ (qy >> 3));
    return h ^ (h >> 16);
}

// The shader compiler folds chains of `seed += .1` inside one iteration of the path loop into base + c_k with
// c_k = fl(c_{k-1} + 0.1f); the base re-materialises once per iteration.
struct Seed {
    float base; int k;
    DEVI float c(int i) const {
        const float c1 = 0.1f, c2 = c1 + 0.1f, c3 = c2 + 0.1f, c4 = c3 + 0.1f;
        return i == 0 ? 0.0f : (i == 1 ? c1 : (i == 2 ? c2 : (i == 3 ? c3 : c4)));
    }
    DEVI uint32_t next_hash() {   // baseHash(floatBitsToUint(vec2(seed+=.1, seed+=.1)))  frag:367,373,380
        const float a = base + c(k + 1), b = base + c(k + 2);
        k += 2;
        return base_hash(f2u(a), f2u(b));
    }
    DEVI void rebase() { base = base + c(k); k = 0; }
};
DEVI float hash1(Seed &s) { return (float)s.next_hash() / 4294967296.0f; }   // frag:366-369 (power-of-two divisor: exact)

// frag:399-404
DEVI f3 random_in_unit_sphere(Seed &s) {
    const uint32_t n = s.next_hash();
    const float h3x = (float)(n & 0x7fffffffu) / 2147483648.0f;
    const float h3y = (float)((n * 16807u) & 0x7fffffffu) / 2147483648.0f;
    const float h3z = (float)((n * 48271u) & 0x7fffffffu) / 2147483648.0f;
    const float hx = h3x * 2.0f - 1.0f;
    const float phi = h3y * 6.28318530718f;
    const float r = pow_poly(h3z, 1.0f / 3.0f);
    const float q = sqrt_ieee(1.0f - hx * hx);
    return F3(r * (q * sincos_poly<true>(phi)), r * (q * sincos_poly<false>(phi)), r * hx);
}
// frag:386-396
DEVI f3 random_cos_weighted_hemisphere_direction(f3 n, Seed &s) {
    const uint32_t h = s.next_hash();
    const float r_x = (float)(h & 0x7fffffffu) / 2147483648.0f;
    const float r_y = (float)((h * 48271u) & 0x7fffffffu) / 2147483648.0f;
    const f3 axis = __builtin_fabsf(n.y) > 0.5f ? F3(1.0f, 0.0f, 0.0f) : F3(0.0f, 1.0f, 0.0f);
    const f3 uu = normalize3(cross3(n, axis));
    const f3 vv = cross3(uu, n);
    const float ra = sqrt_ieee(r_y);
    const float rx = ra * sincos_poly<false>(6.28318530718f * r_x);
    const float ry = ra * sincos_poly<true>(6.28318530718f * r_x);
    const float rz = sqrt_ieee(1.0f - r_y);
    return normalize3((uu * rx + vv * ry) + n * rz);
}
DEVI f3 reflect3(f3 I, f3 N) { const float d = dot3(N, I); return I - N * (2.0f * d); }
DEVI f3 refract3(f3 I, f3 N, float eta) {
    const float ndi = dot3(N, I);
    const float k = 1.0f - eta * (eta * (1.0f - ndi * ndi));
    if (k < 0.0f) return F3(0.0f, 0.0f, 0.0f);
    return I * eta - N * (eta * ndi + sqrt_ieee(k));
}
// frag:412-414
DEVI float fresnel_schlick_roughness(float cos_theta, float F0, float roughness) {
    return F0 + (fmaxf(1.0f - roughness, F0) - F0) * pow_poly(__builtin_fabsf(1.0f - cos_theta), 5.0f);
}

// ------------------------------------------------------------------------------------------------
// scene constants of the shader (frag:95-201)
// ------------------------------------------------------------------------------------------------
struct Material { uint32_t type; float r, g, b, roughness, refractive_index; };
enum { LAMBERTIAN = 0, METAL = 1, DIELECTRIC = 2 };
__constant__ Material c_materials[7] = {
    {METAL, 0.3f, 0.5f, 0.4f, 0.1f, 1.5f},   {METAL, 1.0f, 0.71f, 0.29f, 0.02f, 1.5f}, {LAMBERTIAN, 0.0f, 1.0f, 0.0f, 0.3f, 1.5f},
    {METAL, 1.0f, 1.0f, 0.0f, 0.9f, 1.5f},   {METAL, 1.0f, 0.0f, 0.0f, 0.01f, 1.5f},  {DIELECTRIC, 0.0f, 1.0f, 1.0f, 0.3f, 1.8f},
    {DIELECTRIC, 0.0f, 1.0f, 1.0f, 0.0f, 1.5f},
};
#define NUM_SPHERES 84
// centre xyz + radius; sphere 0 is the ground, 1..80 the small ones (y = r = 0.2), 81..83 the big ones
#define SPHERE_TABLE \
    {0.0f, -1003.0f, 0.0f, 1000.0f}, \
    {-7.995381f, 0.2f, -7.478668f, 0.2f}, {-7.696819f, 0.2f, -5.468978f, 0.2f}, {-7.824804f, 0.2f, -3.120637f, 0.2f}, {-7.132909f, 0.2f, -1.701323f, 0.2f}, \
    {-7.569523f, 0.2f, 0.494554f, 0.2f},  {-7.730332f, 0.2f, 2.358976f, 0.2f},  {-7.892865f, 0.2f, 4.753728f, 0.2f},  {-7.656691f, 0.2f, 6.888913f, 0.2f}, \
    {-7.217835f, 0.2f, 8.203466f, 0.2f},  {-5.115232f, 0.2f, -7.980404f, 0.2f}, {-5.323222f, 0.2f, -5.113037f, 0.2f}, {-5.410681f, 0.2f, -3.527741f, 0.2f}, \
    {-5.460670f, 0.2f, -1.166543f, 0.2f}, {-5.457659f, 0.2f, 0.363870f, 0.2f},  {-5.798715f, 0.2f, 2.161684f, 0.2f},  {-5.116586f, 0.2f, 4.470188f, 0.2f}, \
    {-5.273591f, 0.2f, 6.795187f, 0.2f},  {-5.120286f, 0.2f, 8.731398f, 0.2f},  {-3.601565f, 0.2f, -7.895600f, 0.2f}, {-3.735860f, 0.2f, -5.163056f, 0.2f}, \
    {-3.481116f, 0.2f, -3.794556f, 0.2f}, {-3.866858f, 0.2f, -1.465965f, 0.2f}, {-3.168870f, 0.2f, 0.553099f, 0.2f},  {-3.428552f, 0.2f, 2.627547f, 0.2f}, \
    {-3.771736f, 0.2f, 4.324785f, 0.2f},  {-3.768522f, 0.2f, 6.384588f, 0.2f},  {-3.286992f, 0.2f, 8.441148f, 0.2f},  {-1.552127f, 0.2f, -7.728200f, 0.2f}, \
    {-1.360796f, 0.2f, -5.346098f, 0.2f}, {-1.287209f, 0.2f, -3.735321f, 0.2f}, {-1.344859f, 0.2f, -1.726654f, 0.2f}, {-1.974774f, 0.2f, 0.183260f, 0.2f}, \
    {-1.542872f, 0.2f, 2.067868f, 0.2f},  {-1.743856f, 0.2f, 4.752810f, 0.2f},  {-1.955621f, 0.2f, 6.493702f, 0.2f},  {-1.350449f, 0.2f, 8.068503f, 0.2f}, \
    {0.706123f, 0.2f, -7.116040f, 0.2f},  {0.897766f, 0.2f, -5.938681f, 0.2f},  {0.744113f, 0.2f, -3.402960f, 0.2f},  {0.867750f, 0.2f, -1.311908f, 0.2f}, \
    {0.082480f, 0.2f, 0.838206f, 0.2f},   {0.649692f, 0.2f, 2.525103f, 0.2f},   {0.378574f, 0.2f, 4.055579f, 0.2f},   {0.425844f, 0.2f, 6.098526f, 0.2f}, \
    {0.261365f, 0.2f, 8.661150f, 0.2f},   {2.814218f, 0.2f, -7.751227f, 0.2f},  {2.050073f, 0.2f, -5.731364f, 0.2f},  {2.020130f, 0.2f, -3.472627f, 0.2f}, \
    {2.884277f, 0.2f, -1.232662f, 0.2f},  {2.644454f, 0.2f, 0.596324f, 0.2f},   {2.194283f, 0.2f, 2.880603f, 0.2f},   {2.281000f, 0.2f, 4.094307f, 0.2f}, \
    {2.080841f, 0.2f, 6.716384f, 0.2f},   {2.287131f, 0.2f, 8.583242f, 0.2f},   {4.329136f, 0.2f, -7.497218f, 0.2f},  {4.502115f, 0.2f, -5.941060f, 0.2f}, \
    {4.750631f, 0.2f, -3.836759f, 0.2f},  {4.082084f, 0.2f, -1.180746f, 0.2f},  {4.429173f, 0.2f, 2.069721f, 0.2f},   {4.277152f, 0.2f, 4.297482f, 0.2f}, \
    {4.012743f, 0.2f, 6.225072f, 0.2f},   {4.047066f, 0.2f, 8.419360f, 0.2f},   {6.441846f, 0.2f, -7.700798f, 0.2f},  {6.047810f, 0.2f, -5.519369f, 0.2f}, \
    {6.779211f, 0.2f, -3.740542f, 0.2f},  {6.430776f, 0.2f, -1.332107f, 0.2f},  {6.476387f, 0.2f, 0.329973f, 0.2f},   {6.568686f, 0.2f, 2.116949f, 0.2f}, \
    {6.371189f, 0.2f, 4.609841f, 0.2f},   {6.011877f, 0.2f, 6.569579f, 0.2f},   {6.096087f, 0.2f, 8.892333f, 0.2f},   {8.185763f, 0.2f, -7.191109f, 0.2f}, \
    {8.411960f, 0.2f, -5.285309f, 0.2f},  {8.047109f, 0.2f, -3.427552f, 0.2f},  {8.119639f, 0.2f, -1.652587f, 0.2f},  {8.818120f, 0.2f, 0.401292f, 0.2f}, \
    {8.754155f, 0.2f, 2.152549f, 0.2f},   {8.595298f, 0.2f, 4.802001f, 0.2f},   {8.036216f, 0.2f, 6.739752f, 0.2f},   {8.256561f, 0.2f, 8.129115f, 0.2f}, \
    {0.0f, 2.0f, 0.0f, 1.0f}, {-4.0f, 2.0f, 0.0f, 1.0f}, {4.0f, 2.0f, 0.0f, 1.0f},
__constant__ float4 c_spheres[NUM_SPHERES] = {SPHERE_TABLE};

// Culling structure over the sphere table, built at compile time.  Spheres of radius <= SG_SMALL_RADIUS are binned by
// their footprint, dilated by SG_DILATE on every side, into a grid of SG_CELL x SG_CELL cells over x and z; they all
// lie between two horizontal planes (again dilated).  A ray can only be accepted by the sphere test for such a sphere
// if it passes within radius + (rounding slack) of the centre, and for ray origins within SG_ORIGIN_LIMIT of the world
// origin the rounding slack of the float test (frag:233-252) stays far below SG_DILATE: the discriminant b*b - c
// carries an absolute error below 1e-2 there, while dilating a 0.2 sphere by 0.05 moves it by 2.2e-2.  So testing
// only the spheres binned into the cells that the ray crosses inside the slab (plus every larger sphere), in index
// order, gives exactly the result of the full loop; rays that start farther out take the full loop.
#define SG_CELL 2.0f
#define SG_MAX 12
#define SG_DILATE 0.05f
#define SG_SMALL_RADIUS 0.5f
#define SG_ORIGIN_LIMIT 32.0f
struct SphereGrid {
    float x0, x1, y0, y1, z0, z1;    // dilated bounds of the binned spheres
    int32_t nx, nz;
    uint32_t always[3];              // spheres every ray tests (not binned)
    uint32_t cell[SG_MAX * SG_MAX][3];
};
constexpr int sg_floor(float v) { const int i = (int)v; return (float)i > v ? i - 1 : i; }
constexpr SphereGrid make_sphere_grid() {
    constexpr float tab[NUM_SPHERES][4] = {SPHERE_TABLE};
    SphereGrid g = {};
    bool any = false;
    for (int i = 0; i < NUM_SPHERES; i++) {
        const float r = tab[i][3] + SG_DILATE;
        if (tab[i][3] > SG_SMALL_RADIUS) { g.always[i >> 5] |= 1u << (i & 31); continue; }
        const float lo[3] = {tab[i][0] - r, tab[i][1] - r, tab[i][2] - r}, hi[3] = {tab[i][0] + r, tab[i][1] + r, tab[i][2] + r};
        if (!any) { g.x0 = lo[0]; g.x1 = hi[0]; g.y0 = lo[1]; g.y1 = hi[1]; g.z0 = lo[2]; g.z1 = hi[2]; any = true; }
        g.x0 = lo[0] < g.x0 ? lo[0] : g.x0; g.x1 = hi[0] > g.x1 ? hi[0] : g.x1;
        g.y0 = lo[1] < g.y0 ? lo[1] : g.y0; g.y1 = hi[1] > g.y1 ? hi[1] : g.y1;
        g.z0 = lo[2] < g.z0 ? lo[2] : g.z0; g.z1 = hi[2] > g.z1 ? hi[2] : g.z1;
    }
    g.nx = sg_floor((g.x1 - g.x0) / SG_CELL) + 1;
    g.nz = sg_floor((g.z1 - g.z0) / SG_CELL) + 1;
    const bool fits = g.nx <= SG_MAX && g.nz <= SG_MAX;
    for (int i = 0; i < NUM_SPHERES; i++) {
        if (tab[i][3] > SG_SMALL_RADIUS) continue;
        if (!fits) { g.always[i >> 5] |= 1u << (i & 31); continue; }   // table too wide for the grid: no culling
        const float r = tab[i][3] + SG_DILATE;
        const int xa = sg_floor((tab[i][0] - r - g.x0) / SG_CELL), xb = sg_floor((tab[i][0] + r - g.x0) / SG_CELL);
        const int za = sg_floor((tab[i][2] - r - g.z0) / SG_CELL), zb = sg_floor((tab[i][2] + r - g.z0) / SG_CELL);
        for (int x = xa < 0 ? 0 : xa; x <= xb && x < g.nx; x++)
            for (int z = za < 0 ? 0 : za; z <= zb && z < g.nz; z++) g.cell[x * SG_MAX + z][i >> 5] |= 1u << (i & 31);
    }
    if (!fits) { g.nx = 0; g.nz = 0; }
    return g;
}
__constant__ SphereGrid c_sphere_grid = make_sphere_grid();



// ------------------------------------------------------------------------------------------------
// intersection
// ------------------------------------------------------------------------------------------------
struct Ray { f3 pos, dir, inv; };
struct Hit { bool has_hit; f3 position, normal; uint32_t material_id, portal_id; uint32_t kind, slot; float t; };   // kind / slot / t: what the debug planes of a first segment show (RT_HIT_*, sphere index or triangle slot, distance)
struct Counters { uint32_t rays, pops, tri_tests, box_tests, tri_shaded, entered, sph_tests, lds_pops, fallbacks, cand_checks; };

// frag:417-425
DEVI bool ray_aabb(const Ray &r, f3 lo, f3 hi, float closest) {
    const f3 t1 = (lo - r.pos) * r.inv, t2 = (hi - r.pos) * r.inv;
    const float tmin = fmaxf(fmaxf(fminf(t1.x, t2.x), fminf(t1.y, t2.y)), fminf(t1.z, t2.z));
    const float tmax = fminf(fminf(fmaxf(t1.x, t2.x), fmaxf(t1.y, t2.y)), fmaxf(t1.z, t2.z));
    return tmax > 0.0f && tmin < tmax && tmin < closest;
}
DEVI bool root_box_test(const RtDeviceScene &S, const Ray &r, float closest) {
    if (S.n_nodes == 0) return false;
    return ray_aabb(r, F3(S.bounds_min[0], S.bounds_min[1], S.bounds_min[2]), F3(S.bounds_max[0], S.bounds_max[1], S.bounds_max[2]), closest);
}

// frag:233-252 applied to one sphere given oc = ray.pos - centre and c = dot(oc,oc) - r*r.  A sphere's candidate t
// does not depend on the running closest distance except through the final accept filter, so the reference's
// sequential rule (first index wins ties, frag:439-445) is kept by testing in index order.
DEVI void sphere_accept(float b, float c, uint32_t i, float &closest, uint32_t &closest_i) {
    const float disc = b * b - c;
    if (disc < 0.0f) return;
    const float sq = sqrt_ieee(disc);
    const float t1 = -b - sq, t2 = -b + sq;
    const float t = t1 < 0.0f ? t2 : t1;
    if (t < closest && t > 0.0f) { closest = t; closest_i = i; }
}
// The sphere records are read from a workgroup-shared LDS copy: loads the compiler can keep in flight ahead of the
// arithmetic, whereas scalar loads from constant memory cannot be waited for individually.
struct SphereLds { float4 tab[NUM_SPHERES]; SphereGrid grid; };
// (no barrier of its own: stage_portals, which every kernel calls next, ends with the workgroup barrier for both)
DEVI void stage_spheres(SphereLds *lds, const float4 *src) {
    for (uint32_t i = threadIdx.x; i < NUM_SPHERES; i += blockDim.x) lds->tab[i] = src[i];
    uint32_t *dst = reinterpret_cast<uint32_t *>(&lds->grid);
    const uint32_t *g = reinterpret_cast<const uint32_t *>(&c_sphere_grid);
    for (uint32_t i = threadIdx.x; i < sizeof(SphereGrid) / 4u; i += blockDim.x) dst[i] = g[i];
}
// Set of spheres that the ray can possibly be accepted by (a superset is always correct).
struct SphereMask { uint32_t w[3]; };
DEVI SphereMask sphere_candidates(const SphereGrid *g, const Ray &r) {
    SphereMask m = {{g->always[0], g->always[1], g->always[2]}};
    const SphereMask all = {{0xFFFFFFFFu, 0xFFFFFFFFu, (1u << (NUM_SPHERES - 64)) - 1u}};
    if (g->nx == 0) return m;
    const float far = fmaxf(fmaxf(fabsf(r.pos.x), fabsf(r.pos.y)), fabsf(r.pos.z));
    if (!(far <= SG_ORIGIN_LIMIT)) return all;
    // part of the ray inside the slab of the binned spheres (a NaN from 0 * inf drops out of min/max)
    const float tx1 = (g->x0 - r.pos.x) * r.inv.x, tx2 = (g->x1 - r.pos.x) * r.inv.x;
    const float ty1 = (g->y0 - r.pos.y) * r.inv.y, ty2 = (g->y1 - r.pos.y) * r.inv.y;
    const float tz1 = (g->z0 - r.pos.z) * r.inv.z, tz2 = (g->z1 - r.pos.z) * r.inv.z;
    const float t0 = fmaxf(fmaxf(fminf(tx1, tx2), fminf(ty1, ty2)), fmaxf(fminf(tz1, tz2), 0.0f));
    const float t1 = fminf(fminf(fmaxf(tx1, tx2), fmaxf(ty1, ty2)), fmaxf(tz1, tz2));
    if (!(t1 < 1e30f)) return all;     // not a proper ray (NaN or zero direction): no culling
    if (!(t0 <= t1)) return m;
    const float ax = r.pos.x + r.dir.x * t0, bx = r.pos.x + r.dir.x * t1;
    const float az = r.pos.z + r.dir.z * t0, bz = r.pos.z + r.dir.z * t1;
    const float ic = 1.0f / SG_CELL, slack = 1e-3f;
    const int nx1 = g->nx - 1, nz1 = g->nz - 1;
    const int ixa = min(max((int)floorf((fminf(ax, bx) - g->x0) * ic - slack), 0), nx1);
    const int ixb = min(max((int)floorf((fmaxf(ax, bx) - g->x0) * ic + slack), 0), nx1);
    for (int ix = ixa; ix <= ixb; ix++) {
        float za = az, zb = bz;
        if (ixa != ixb) {   // the ray's stretch inside this column of cells
            const float xl = g->x0 + SG_CELL * (float)ix - 2.0f * slack, xr = xl + SG_CELL + 4.0f * slack;
            const float ta = (xl - r.pos.x) * r.inv.x, tb = (xr - r.pos.x) * r.inv.x;
            const float ts = fminf(fmaxf(fminf(ta, tb), t0), t1), te = fmaxf(fminf(fmaxf(ta, tb), t1), t0);
            za = r.pos.z + r.dir.z * ts; zb = r.pos.z + r.dir.z * te;
        }
        const int iza = min(max((int)floorf((fminf(za, zb) - g->z0) * ic - slack), 0), nz1);
        const int izb = min(max((int)floorf((fmaxf(za, zb) - g->z0) * ic + slack), 0), nz1);
        for (int iz = iza; iz <= izb; iz++) {
            const uint32_t *c = g->cell[ix * SG_MAX + iz];
            m.w[0] |= c[0]; m.w[1] |= c[1]; m.w[2] |= c[2];
        }
    }
    return m;
}
// The sphere loop of frag:439-445 over the candidate set, in index order.  PRIMARY: tab holds (oc.xyz, c) of a shared
// ray origin; otherwise (centre.xyz, radius).
template <bool PRIMARY>
DEVI void intersect_spheres_culled(const SphereLds *lds, const Ray &r, float &closest, uint32_t &closest_i, uint32_t *n_tests = nullptr) {
    const SphereMask m = sphere_candidates(&lds->grid, r);
    if (n_tests) *n_tests += (uint32_t)(__popc(m.w[0]) + __popc(m.w[1]) + __popc(m.w[2]));   // sphere tests actually executed
#pragma unroll
    for (uint32_t w = 0; w < 3; w++) {
        uint32_t bits = m.w[w];
        while (bits) {
            const uint32_t i = w * 32u + (uint32_t)__builtin_ctz(bits);
            bits &= bits - 1u;
            const float4 s = lds->tab[i];
            if (PRIMARY) {
                sphere_accept(dot3(F3(s.x, s.y, s.z), r.dir), s.w, i, closest, closest_i);
            } else {
                const f3 oc = r.pos - F3(s.x, s.y, s.z);
                sphere_accept(dot3(oc, r.dir), dot3(oc, oc) - s.w * s.w, i, closest, closest_i);
            }
        }
    }
}
// Generic origin: lds_tab holds (centre.xyz, radius).
[[maybe_unused]] DEVI void intersect_spheres(const float4 *lds_tab, const Ray &r, float &closest, uint32_t &closest_i) {
#pragma unroll 6
    for (uint32_t i = 0; i < NUM_SPHERES; i++) {
        const float4 s = lds_tab[i];
        const f3 oc = r.pos - F3(s.x, s.y, s.z);
        const float b = dot3(oc, r.dir);
        const float c = dot3(oc, oc) - s.w * s.w;
        sphere_accept(b, c, i, closest, closest_i);
    }
}
// Primary rays share their origin, so oc and c are per-frame constants of each sphere: lds_tab holds (oc.xyz, c)
// computed once with the very same operations.
[[maybe_unused]] DEVI void intersect_spheres_primary(const float4 *lds_tab, f3 dir, float &closest, uint32_t &closest_i) {
#pragma unroll 6
    for (uint32_t i = 0; i < NUM_SPHERES; i++) {
        const float4 s = lds_tab[i];
        const float b = dot3(F3(s.x, s.y, s.z), dir);
        sphere_accept(b, s.w, i, closest, closest_i);
    }
}

// frag:255-281 on a prepared triangle record (first vertex and the two edges)
struct TriData { float4 q0, q1; float e2z; uint32_t id; };   // id: RtTriRec::dup_first, which identifies the triangle behind the reference
DEVI TriData load_triangle(const RtTriRec *tr) {
    TriData d;
    d.q0 = *reinterpret_cast<const float4 *>(tr);
    d.q1 = *(reinterpret_cast<const float4 *>(tr) + 1);
    const uint2 q2 = *reinterpret_cast<const uint2 *>(reinterpret_cast<const float4 *>(tr) + 2);
    d.e2z = u2f(q2.x); d.id = q2.y;
    return d;
}
// Keeps the loads of a record where they are written.  Without it the compiler moves the load of the first vertex into the
// branch of the intersection test that needs it (behind the determinant check), where it costs a second memory round trip per test.
DEVI void pin_triangle(TriData &d) {
    asm volatile("" : "+v"(d.q0.x), "+v"(d.q0.y), "+v"(d.q0.z), "+v"(d.q0.w), "+v"(d.q1.x), "+v"(d.q1.y), "+v"(d.q1.z), "+v"(d.q1.w), "+v"(d.e2z), "+v"(d.id));
}
DEVI float ray_triangle(const Ray &r, const TriData &d) {
    const f3 v1 = F3(d.q0.x, d.q0.y, d.q0.z), e1 = F3(d.q0.w, d.q1.x, d.q1.y), e2 = F3(d.q1.z, d.q1.w, d.e2z);
    const f3 pvec = cross3(r.dir, e2);
    const float det = dot3(e1, pvec);
    if (__builtin_fabsf(det) < 0.000000000000001f) return -1.0f;
    const float inv_det = rcp_ieee(det);
    const f3 tvec = r.pos - v1;
    const float ua = dot3(tvec, pvec);
    const float u = inv_det * ua;
    if (u < 0.0f || u > 1.0f) return -1.0f;
    const f3 qvec = cross3(tvec, e1);
    const float va = dot3(r.dir, qvec);
    const float v = inv_det * va;
    // u + v of frag:277 is a sum of two products with the common factor invDet, which the shader compiler the reference runs under
    // factors out: invDet * (ua + va).  An ulp apart from u + v in a quarter of the cases, which decides hits on the triangle's third edge.
    if (v < 0.0f || inv_det * (ua + va) > 1.0f) return -1.0f;
    return dot3(e2, qvec) * inv_det;
}

// Depth-first traversal state (frag:431-433,455-514).  The reference keeps a LIFO of (box, node) pairs: a visit
// pushes the children whose boxes pass the slab test in ascending octant order and the next visit pops the top, so the
// passing children of a node are visited in descending octant order, each with its whole subtree, before the
// node's siblings.  That order is reproduced with ONE entry per tree level instead of one per pending node: an entry
// holds the record index of the node's block of children and the set of passing children still to visit, as a
// bitmask over octants (entry = first_child / 8 | octant_mask << 24: blocks of children start at multiples of eight; child o is
// record first_child + o).  The slab tests happen, as
// in the reference, when the parent is visited -- against the closest distance known at that moment -- and are
// never repeated.  The stack is a column of LDS: entry k of thread t lives at lds[k * threads + t] (no bank conflicts
// whatever the lanes' depths), and it can never hold more than one entry per tree level.
template <int STRIDE>   // threads per workgroup = distance in words between two entries of one lane
struct LevelStackT {
    uint32_t *col;   // &lds[threadIdx.x]
    int sp;
    // the root is octant 0 of a virtual level; `key` is the octant permutation of this ray (0: the reference's order)
    DEVI void reset_to_root(uint32_t key) { col[0] = 0u | ((1u << key) << 24); sp = 1; }
    DEVI bool empty() const { return sp == 0; }
    DEVI uint32_t pop_node(uint32_t key) {
        const uint32_t e = col[(sp - 1) * STRIDE];
        const uint32_t mask = e >> 24;
        const uint32_t r = 31u - (uint32_t)__clz((int)mask);      // highest pending octant
        const uint32_t rest = mask & ~(1u << r);
        if (rest) col[(sp - 1) * STRIDE] = (e & 0x00FFFFFFu) | (rest << 24);
        else sp--;
        return ((e & 0x00FFFFFFu) << 3) + (r ^ key);
    }
    DEVI void push_level(uint32_t first_child, uint32_t octant_mask) { col[sp * STRIDE] = (first_child >> 3) | (octant_mask << 24); sp++; }
};
typedef LevelStackT<256> LevelStack;

// Slab tests of the up-to-eight child boxes of a node (frag:489-513).  The eight boxes only use three plane
// positions per axis (lo, mid, hi), so the reference's per-child (bound - origin) * inverse_direction products and
// their min/max are computed once per plane pair and shared: every value is produced by the same operation on the same
// operands as in the reference, only not recomputed.  Returns the existing children that pass as a bitmask over octants.
// `near_tie` (exact near-first mode only, else null): set when the entry distance of any child box lies within `band` of the
// closest distance on either side -- a pruning decision that rounding could have turned the other way, or one that the
// reference's strict comparison and an inclusive one would make differently (see Exact).  Every child's entry distance is one
// of the six per-axis values below, so those are what is compared.
#ifndef EXACT_BAND
#define EXACT_BAND 0x1p-18f
#endif
DEVI float exact_band(float closest) { return fmaxf(closest * EXACT_BAND, 0x1p-126f); }   // (infinite for an infinite distance: nothing is within it, see below)
// max / min of three as ONE instruction.  Written as nested fmaxf / fminf the compiler shares the inner results between the eight
// children of a node (four inner and eight outer operations per side instead of eight three-operand ones); same NaN rule as the
// nested form (an operand that is NaN drops out).
DEVI float max3f(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
DEVI float min3f(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// FOLD: the closest distance is folded into the far side of the z slab, so that a child costs ONE comparison instead of three:
// with m = min(tmax, closest), `tmax > 0 && tmin < tmax && tmin < closest` is `m > 0 && tmin < m` whenever closest > 0 (a NaN
// plane drops out of either form alike).  With a closest distance of exactly 0 the folded form passes no child at all where the
// reference still enters boxes around the ray's origin -- but nothing can be accepted after a hit at distance 0 (0 <= t < 0), so
// the outcome of the traversal is the same; only the work counters would see the difference, and the counting instantiations do
// not fold.  (The near-first pass of the exact order, which looks for other triangles at the closest distance, gives a ray with a
// hit at distance 0 to the reference's order: accept_hit.)
template <bool BAND = false, bool FOLD = false>
DEVI uint32_t child_pass_octants(const Ray &r, f3 lo, f3 mid, f3 hi, uint32_t child_mask, float closest, float band = 0.0f, bool *near_tie = nullptr) {
    const f3 tl = (lo - r.pos) * r.inv, tm = (mid - r.pos) * r.inv, th = (hi - r.pos) * r.inv;
    // lower half of an axis: box [lo, mid] -> t1 = tl, t2 = tm; upper half: box [mid, hi] -> t1 = tm, t2 = th
    const f3 near0 = F3(fminf(tl.x, tm.x), fminf(tl.y, tm.y), fminf(tl.z, tm.z)), far0 = F3(fmaxf(tl.x, tm.x), fmaxf(tl.y, tm.y), fmaxf(tl.z, tm.z));
    const f3 near1 = F3(fminf(tm.x, th.x), fminf(tm.y, th.y), fminf(tm.z, th.z)), far1 = F3(fmaxf(tm.x, th.x), fmaxf(tm.y, th.y), fmaxf(tm.z, th.z));
    if (BAND) {
        // Straight-line code on purpose (the caller passes a band of zero where the flag does not apply): a branch here would end the
        // basic block between the products above and the minima / maxima below, and the compiler then re-canonicalises all eighteen
        // products before using them (fourteen more instructions per visit).
        // (NaN differences -- 0 * inf planes -- drop out of the minima; with an infinite closest distance every difference is
        // infinite and not BELOW the infinite band)
        const float d0 = fminf(fminf(__builtin_fabsf(near0.x - closest), __builtin_fabsf(near0.y - closest)), __builtin_fabsf(near0.z - closest));
        const float d1 = fminf(fminf(__builtin_fabsf(near1.x - closest), __builtin_fabsf(near1.y - closest)), __builtin_fabsf(near1.z - closest));
        *near_tie = *near_tie | (fminf(d0, d1) < band);
    }
    uint32_t oct_pass = 0;
    if (FOLD) {
        // ... and 0 into the near side: max(tmin, 0) < m is `tmin < m && 0 < m`, one comparison per child.  (The rays that get here
        // have passed the root box test, so they are not NaN rays; a NaN plane of one axis drops out of either form.)
        const float fz0 = fminf(far0.z, closest), fz1 = fminf(far1.z, closest);
        const float nz0 = fmaxf(near0.z, 0.0f), nz1 = fmaxf(near1.z, 0.0f);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float nx = (i & 1) ? near1.x : near0.x, ny = (i & 2) ? near1.y : near0.y, nz = (i & 4) ? nz1 : nz0;
            const float fx = (i & 1) ? far1.x : far0.x, fy = (i & 2) ? far1.y : far0.y, fz = (i & 4) ? fz1 : fz0;
            if (max3f(nx, ny, nz) < min3f(fx, fy, fz)) oct_pass |= 1u << i;
        }
        return oct_pass & child_mask;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const float nx = (i & 1) ? near1.x : near0.x, ny = (i & 2) ? near1.y : near0.y, nz = (i & 4) ? near1.z : near0.z;
        const float fx = (i & 1) ? far1.x : far0.x, fy = (i & 2) ? far1.y : far0.y, fz = (i & 4) ? far1.z : far0.z;
        const float tmin = max3f(nx, ny, nz), tmax = min3f(fx, fy, fz);
        if (tmax > 0.0f && tmin < tmax && tmin < closest) oct_pass |= 1u << i;
    }
    return oct_pass & child_mask;
}

struct NodeData { uint32_t tri_start, tri_count, child_mask, first_child, rank, first_id; f3 lo, mid, hi; };   // first_id: a leaf's first triangle (RtNodeRec::first_id)
DEVI NodeData unpack_node(const uint4 w0, const float4 w1, const float4 w2, float hiz, uint32_t first_id) {
    NodeData n;
    n.tri_start = w0.y; n.tri_count = w0.x >> 16; n.child_mask = (w0.x >> 8) & 0xffu; n.first_child = w0.z; n.rank = w0.w; n.first_id = first_id;
    n.lo = F3(w1.x, w1.y, w1.z); n.mid = F3(w1.w, w2.x, w2.y); n.hi = F3(w2.z, w2.w, hiz);
    return n;
}
DEVI NodeData load_node(const RtDeviceScene &S, uint32_t ni) {
    const uint4 *rec = reinterpret_cast<const uint4 *>(S.nodes + ni);
    const uint4 w3 = rec[3];     // hi.z, first_id, parent, pad
    return unpack_node(rec[0], reinterpret_cast<const float4 *>(rec)[1], reinterpret_cast<const float4 *>(rec)[2], u2f(w3.x), w3.y);
}
// The records of the top three levels of the tree (the first RT_TOP_RECORDS records: the root's block, its children's block and
// their blocks), which every ray that enters the octree visits, staged once per workgroup in LDS.
// Beside them, for the exact near-first order, the records of the scene's face-coplanar triangles (RtDeviceScene::risky_tris), which
// every ray is tested against before it starts.
struct TopLevels { uint4 q[RT_TOP_RECORDS][4]; uint4 risky[RT_MAX_RISKY][3]; };
DEVI void stage_top_levels(TopLevels *top, const RtDeviceScene &S, bool with_risky) {      // the caller's next barrier publishes the copy
    const uint32_t n = min(S.n_nodes, RT_TOP_RECORDS) * 4u;
    const uint4 *src = reinterpret_cast<const uint4 *>(S.nodes);
    uint4 *dst = &top->q[0][0];
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
    if (with_risky && S.n_risky <= RT_MAX_RISKY) {
        const uint4 *rs = reinterpret_cast<const uint4 *>(S.risky_tris);
        for (uint32_t i = threadIdx.x; i < S.n_risky * 3u; i += blockDim.x) top->risky[i / 3u][i % 3u] = rs[i];
    }
}
template <bool COUNT>
DEVI NodeData load_node(const RtDeviceScene &S, const TopLevels *top, uint32_t ni, Counters &cn) {
    if (top && ni < RT_TOP_RECORDS) {
        if (COUNT) cn.lds_pops++;
        const uint4 *rec = top->q[ni];
        const uint4 a = rec[0], b = rec[1], c = rec[2], d = rec[3];
        return unpack_node(a, make_float4(u2f(b.x), u2f(b.y), u2f(b.z), u2f(b.w)), make_float4(u2f(c.x), u2f(c.y), u2f(c.z), u2f(c.w)), u2f(d.x), d.y);
    }
    return load_node(S, ni);
}

// The triangle record a leaf carries in place of its box (lo = v1, mid = e1, hi = e2).
DEVI TriData inline_triangle(const NodeData &n) {
    TriData d;
    d.q0 = make_float4(n.lo.x, n.lo.y, n.lo.z, n.mid.x);
    d.q1 = make_float4(n.mid.y, n.mid.z, n.hi.x, n.hi.y);
    d.e2z = n.hi.z; d.id = n.first_id;
    return d;
}

// Order in which the passing children of a node are visited.  The reference pushes them in ascending octant order and pops
// the highest first, whatever the ray's direction (frag:489-513).  Optionally (RtLaunch::near_first) they are visited
// nearest first: the level entry then stores the mask with bit (octant ^ key), key = ~(sign bits of the direction), so that
// popping the highest bit yields the octant on the ray's near side first.  Closer hits are then found earlier and later
// box tests reject more (the bunny scene: 5 % fewer visits, the 100 k scene: 60 %), but the set of triangles tested is
// no longer the reference's, so equality of the results is no longer guaranteed bit for bit (see rt_hotpath.h).
DEVI uint32_t order_key(const Ray &r, bool near_first) {
    return near_first ? (((r.dir.x < 0.0f ? 1u : 0u) | (r.dir.y < 0.0f ? 2u : 0u) | (r.dir.z < 0.0f ? 4u : 0u)) ^ 7u) : 0u;
}
DEVI uint32_t permute_octants(uint32_t mask, uint32_t key) {      // bit o -> bit (o ^ key)
    if (key & 1u) mask = ((mask & 0x55u) << 1) | ((mask & 0xAAu) >> 1);
    if (key & 2u) mask = ((mask & 0x33u) << 2) | ((mask & 0xCCu) >> 2);
    if (key & 4u) mask = ((mask & 0x0Fu) << 4) | ((mask & 0xF0u) >> 4);
    return mask;
}
#ifndef TRI_BATCH
#define TRI_BATCH 2u   // more in flight costs registers, and with them resident waves
#endif
// One visit (frag:456-513) by a single lane: test the node's triangles in order, then the boxes of its children.
template <bool COUNT, class STACK>
DEVI void visit_node(const RtDeviceScene &S, const Ray &r, STACK &st, uint32_t key, bool near_first, float &closest, uint32_t &closest_tri, bool &tri_hit, Counters &cn) {
    const NodeData n = load_node(S, st.pop_node(key));
    if (COUNT) { cn.pops++; cn.tri_tests += n.tri_count; cn.box_tests += __popc(n.child_mask); }
    // triangle records are fetched TRI_BATCH at a time so that their memory latencies overlap; they are tested in order
    // (the copy of a leaf's first triangle inside its node record is not used here: the extra branch per visit cost this
    // kernel more than the saved round trip gave; walk_step uses it)
    for (uint32_t i0 = 0; i0 < n.tri_count; i0 += TRI_BATCH) {
        TriData td[TRI_BATCH];
#pragma unroll
        for (uint32_t k = 0; k < TRI_BATCH; k++) td[k] = load_triangle(S.tris + (n.tri_start + min(i0 + k, n.tri_count - 1u)));
#pragma unroll
        for (uint32_t k = 0; k < TRI_BATCH; k++) {
            if (i0 + k < n.tri_count) {
                const float t = ray_triangle(r, td[k]);
                if (t >= 0.0f && t < closest) { closest = t; closest_tri = n.tri_start + i0 + k; tri_hit = true; }
            }
        }
    }
    if (n.child_mask == 0) return;
    uint32_t pass = child_pass_octants<false, !COUNT>(r, n.lo, n.mid, n.hi, n.child_mask, closest);
    if (near_first) pass = permute_octants(pass, key);      // (uniform branch)
    if (pass) st.push_level(n.first_child, pass);
}

// The same traversal as a wave-wide state machine.  A lane's traversal alternates between two kinds of step: testing the
// next TRI_BATCH triangles of its current node, and finishing the node (child boxes against the closest distance the
// triangles left, push) followed by popping and loading the next one.  With one lane per visit (visit_node) a wave runs
// the triangle loop for its lanes at leaves while its lanes at inner nodes wait, and the box tests the other way round;
// here a wave runs, in every step, the kind of work the majority of its traversing lanes is waiting for.  Per lane the
// order of triangle tests, box tests, pushes and pops is exactly visit_node's.
struct Walk {
    NodeData node;            // current node: its child boxes are still to be tested while node.child_mask != 0
    uint32_t tri_i, tri_end;  // its triangles still to be tested
    uint32_t key;             // octant permutation of this ray (order_key)
    template <class STACK> DEVI void begin(STACK &st, uint32_t k) { key = k; st.reset_to_root(k); node.child_mask = 0; tri_i = tri_end = 0; }
};

// Exact near-first traversal (RtLaunch::near_first == 2): the children nearest first, and still the reference shader's
// result -- the closest distance AND the triangle reference ("slot") it reports -- in three steps per ray.
//
// The reference (R) visits a node's passing children in descending octant order, tests `t < closest` strictly, and prunes a
// child whose box is entered at tmin >= the closest distance known when its parent is visited (frag:456-513).  Its slot is
// therefore the FIRST one, in its visiting order, among the references with the final distance t* in the nodes it visits.
//
// Both traversals can only find hits in the same set H: references held by nodes whose boxes -- and all their ancestors' -- pass
// the part of the slab test that does not depend on the closest distance; box entry distances never decrease from a node to its
// children (the same rounded operations on nested planes).  If the nearest hit of H has a holder that is entered no later than
// the hit (tmin <= t for the node), then R and any other pruning order end with that distance: nothing can prune the chain of
// boxes leading to it before it is found.
//
//  1. Near-first pass.  Children in the order of the ray's direction, pruning and accepting exactly like R (tmin < closest,
//     t < closest).  Among the references of ONE triangle with equal t that it tests it keeps the one whose node has the lowest
//     rank (RtNodeRec::rank: R's visiting order).  It raises `tie` -- and the ray is traversed again in R's order, step 3 -- when
//       a. a DIFFERENT triangle is hit at the closest distance or within EXACT_BAND (relative) of it, on either side: R's
//          order decides which of the two it keeps, and whether it prunes the box of the other;
//       b. the box of a child is entered within the band of the closest distance, on either side: the comparison that prunes
//          it can fall the other way in R, whose closest distance at that moment may be another one within the band, and a
//          reference with the distance t* in a box entered exactly AT t* is met by neither traversal or by R only.
//     A flag raised against a closest distance that is later improved by more than the band is dropped again.
//     Together a. and b. cover every ray for which the premise above can fail by less than the band: the nearest hit lies up to
//     EXACT_BAND in FRONT of every box that lists its triangle (a hit point on a box face where the build's float triangle/box
//     test left the triangle out of the box the point is in, or where the slab test puts the entry an ulp behind the hit),
//     and R reports something else only if another hit or its pruning comparison falls into that gap.
//  2. Candidate check.  A reference can also give t* from a node entered BEHIND the hit point (tmin > t*: the triangle sticks
//     out of that node's box towards the ray origin).  Whether R visits such a node depends on everything R has met before it.
//     So the result is `uncertain` if the kept reference's own node is entered behind the hit, or if the ray passes through
//     the box (tmax > 0 && tmin < tmax) of a node holding one of the references of the same triangle that R would meet
//     earlier -- the positions [dup_first, dup_pos) of the sorted reference array (RtTriRec); such a node, not tested in
//     step 1, is necessarily one entered behind the hit.
//  3. Uncertain or tie: the lane runs the reference's own traversal from the start (fallback), whose result is R's by definition.
//
// Which of a triangle's references is reported can only be observed through the optional plane rt_target.hit_slot of the
// primary hit: every reference of a triangle holds the same vertex indices and material, so distance, normal, colour and
// depth do not depend on it.  Step 2 therefore runs only for segments whose slot is written out (Exact::slot); all others
// finish after step 1 unless `tie` is set.
//
// A triangle lying exactly IN a face plane of the boxes around it (RtDeviceScene::risky_tris: e.g. the triangles on the faces of
// the root box, whose bounds are the extreme vertices) is listed by the build's float test in some of those boxes only, and a ray
// can hit it well before entering any node that holds it -- beyond any band.  A ray that hits one of these (few) triangles is
// therefore traversed in the reference's order from the start (hits_risky); a scene with more than RT_MAX_RISKY of them does not
// use this mode at all.
//
// When step 1 finds no triangle closer than the initial distance, R -- which prunes at least as much -- finds none either.
//
// The premise fails outright when a triangle test returns a distance that has nothing to do with where the triangle is: a ray that
// lies IN the plane of a triangle has a Moeller-Trumbore determinant of rounding noise, and the test then accepts the triangle --
// seven units away in the record that showed it -- at distance 0 or at any other.  R reports that if it happens to test the
// triangle before a nearer hit prunes its cell; step 1 never makes the test.  For primary rays, which all start at the camera
// position, the triangles this can happen with are found per frame and such rays take R's order (hits_eye_coplanar below).
//
// What is NOT covered by construction: a hit whose computed distance lies MORE than the band in front of every box that lists its
// triangle, with another hit inside that stretch -- a triangle that is not face-coplanar whose cell does not list it where the hit
// point is, the ray running on along that face for longer than the band; or a secondary ray whose origin AND direction lie in the
// plane of another triangle to within rounding.  The llvmpipe records of a scene built for this (sheet edges and corners exactly on
// cell planes, rays inside cell planes, eye and target on the lattice: tests/golden/refscene_planes_*.npz) and the randomised
// comparisons with the CPU restatement run in this mode.
struct Exact {
    uint32_t best_rank;       // rank of the node that holds closest_tri
    uint32_t best_id;         // the triangle behind closest_tri (RtTriRec::dup_first)
    uint32_t post_q, post_end;// candidate positions still to be checked (step 2) while post_q < post_end
    float closest0;           // closest distance before the octree (for the fallback)
    float closest_hi;         // the closest distance times (1 + EXACT_BAND): upper end of the band in which another triangle's distance is a near-tie
    float band;               // exact_band(closest distance): half-width of the band around it in which a box entry is a near-tie
    bool nf, tie;             // near-first pass running (cleared for the fallback); another triangle at the bit-equal distance
    bool slot;                // the reference's SLOT is wanted (step 2), not only its distance and triangle
    DEVI void begin(float c0, bool want_slot) { best_rank = 0; best_id = 0; post_q = post_end = 0; closest0 = c0; closest_hi = c0; band = exact_band(c0); nf = true; tie = false; slot = want_slot; }
};
// Eye-coplanar triangles.  Every primary ray starts at the camera position.  When that point lies IN the plane of a triangle, a ray
// whose direction lies in that plane too gives the Moeller-Trumbore test a determinant and a `u` numerator that are both rounding
// noise (and a `t` numerator of exactly zero when the coordinates are exact): the test then accepts or rejects the triangle at
// random, at a distance that has nothing to do with where the triangle is.  The reference reports such a hit if it happens to
// test the triangle before a nearer hit prunes its cell, so the result depends on the visiting order alone and the near-first pass
// cannot know about it (the cell is pruned, the test never made).  Seen with eye, target and triangle vertices on one lattice.
// Per frame, eye_planes_kernel lists the triangles whose plane passes within 2^-18 of the camera position (relative to the
// distances involved: sixteen times the rounding noise of the quantities that decide the test); a primary ray whose direction
// lies within 2^-10 of the plane of a listed triangle AND whose own test of that triangle -- the reference's arithmetic on the
// reference's operands -- accepts it takes the reference's order from the start, as after hits_risky.  More than RT_EYE_CAP
// listed triangles: every primary ray of the frame does.
__global__ __launch_bounds__(256) void eye_planes_kernel(const RtPlaneRec *__restrict__ planes, uint32_t n, float ex, float ey, float ez, uint32_t *__restrict__ count, float4 *__restrict__ list) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 a = reinterpret_cast<const float4 *>(planes)[2 * i], b = reinterpret_cast<const float4 *>(planes)[2 * i + 1];
    const float tx = ex - a.w, ty = ey - b.x, tz = ez - b.y;
    const float dist = a.x * tx + a.y * ty + a.z * tz;
    const float scale = __builtin_fabsf(tx) + __builtin_fabsf(ty) + __builtin_fabsf(tz) + b.w;
    if (!(__builtin_fabsf(dist) <= scale * 0x1p-18f)) return;
    const uint32_t k = atomicAdd(count, 1u);
    if (k < RT_EYE_CAP) list[k] = make_float4(a.x, a.y, a.z, b.z);     // (b.z: the slot's bits)
}
DEVI bool hits_eye_coplanar(const RtDeviceScene &S, const RtLaunch &L, const Ray &r) {
    if (!L.eye_count) return false;
    const uint32_t n = *L.eye_count;                   // (uniform: the common case is zero)
    if (n == 0) return false;
    if (n > RT_EYE_CAP) return true;
    bool hit = false;
    for (uint32_t k = 0; k < n; k++) {
        const float4 e = L.eye_list[k];
        if (__builtin_fabsf(r.dir.x * e.x + r.dir.y * e.y + r.dir.z * e.z) <= 0x1p-10f) {
            const TriData d = load_triangle(S.tris + f2u(e.w));
            hit = hit || ray_triangle(r, d) >= 0.0f;
        }
    }
    return hit;
}
DEVI bool hits_risky(const RtDeviceScene &S, const TopLevels *top, const Ray &r) {
    bool hit = false;
    for (uint32_t k = 0; k < S.n_risky; k++) {
        const uint4 a = top->risky[k][0], b = top->risky[k][1], c = top->risky[k][2];
        TriData d;
        d.q0 = make_float4(u2f(a.x), u2f(a.y), u2f(a.z), u2f(a.w));
        d.q1 = make_float4(u2f(b.x), u2f(b.y), u2f(b.z), u2f(b.w));
        d.e2z = u2f(c.x); d.id = c.y;
        hit = hit || ray_triangle(r, d) >= 0.0f;
    }
    return hit;
}
// EXACT: 0 = plain traversal; 1 = exact near-first for segments whose slot nobody observes (the tie flag only);
// 2 = the same with the rank bookkeeping that step 2 needs.  (Kept apart because the bookkeeping is not free: on the 1.28 M-triangle
// frame the tie flag costs 0.85 ms of 17.4 and the ranks another 0.6.)
template <int EXACT>
DEVI void accept_hit(float t, uint32_t slot, uint32_t id, uint32_t node_rank, float &closest, uint32_t &closest_tri, bool &tri_hit, Exact &x) {
    if (t >= 0.0f && t < closest) {
        if (EXACT) {
            // a flag raised against the previous closest distance stays only if the new one is within the band of it; the triangle
            // that gave the previous distance (another one: references of one triangle give the bit-equal distance) raises it too
            // (a hit at distance 0 exactly: references of other triangles at the same distance sit in boxes around the ray's origin, which
            // the folded child test -- see child_pass_octants -- no longer enters: such a ray takes the reference's order)
            const float t_hi = t * (1.0f + EXACT_BAND);
            x.tie = (closest <= t_hi && (x.tie || tri_hit)) || t == 0.0f;
            x.closest_hi = t_hi;
            x.band = exact_band(t);
            x.best_id = id;
        }
        closest = t; closest_tri = slot; tri_hit = true;
        if (EXACT == 2) x.best_rank = node_rank;
    } else if (EXACT && tri_hit && t >= closest && t <= x.closest_hi && x.nf) {
        if (id != x.best_id) x.tie = true;
        else if (EXACT == 2 && t == closest && (node_rank < x.best_rank || (node_rank == x.best_rank && slot < closest_tri))) { closest_tri = slot; x.best_rank = node_rank; }
    }
}
// Must be called by all lanes of a wave together; `active` marks the lanes with a live traversal and is cleared when a
// lane's traversal is complete.  mode: RtLaunch::near_first (0: the reference's order, 1: near first, 2: exact near first).
template <bool COUNT, int EXACT, class STACK>
DEVI void walk_step(const RtDeviceScene &S, const TopLevels *top, const Ray &r, STACK &st, Walk &w, Exact &x, bool near_first, float &closest, uint32_t &closest_tri, bool &tri_hit, bool &active, Counters &cn) {
    const bool want_tri = active && w.tri_i < w.tri_end;
    const uint32_t n_tri = (uint32_t)__popcll(__ballot(want_tri)), n_pop = (uint32_t)__popcll(__ballot(active && !want_tri));
    if (n_tri != 0 && n_tri >= n_pop) {
        if (want_tri) {
            TriData td[TRI_BATCH];
#pragma unroll
            for (uint32_t k = 0; k < TRI_BATCH; k++) td[k] = load_triangle(S.tris + min(w.tri_i + k, w.tri_end - 1u));
#pragma unroll
            for (uint32_t k = 0; k < TRI_BATCH; k++) pin_triangle(td[k]);
#pragma unroll
            for (uint32_t k = 0; k < TRI_BATCH; k++) {
                if (w.tri_i + k < w.tri_end) accept_hit<EXACT>(ray_triangle(r, td[k]), w.tri_i + k, td[k].id, w.node.rank, closest, closest_tri, tri_hit, x);
            }
            w.tri_i = min(w.tri_i + TRI_BATCH, w.tri_end);
        }
    } else if (active && !want_tri) {
        if (EXACT == 2 && x.post_q < x.post_end) {
            // step 2, one candidate per step: the box of the node that holds a reference of the hit triangle, derived from its
            // parent's record.  For the kept reference itself (the last position) the question is whether its node is entered
            // no later than the hit (tmin <= t*: then R visits it); for an earlier reference, whether the ray passes through
            // its node at all (then R may have visited it).
            const uint32_t po = S.dup_parent[x.post_q];
            const bool self = x.post_q + 1u == x.post_end;
            bool uncertain = false;
            if (po != RT_NO_PARENT) {            // (a reference held by the root: always visited, by both traversals)
                const NodeData pn = load_node<false>(S, top, RT_PARENT_INDEX(po), cn);
                const uint32_t o = RT_PARENT_OCTANT(po);
                const f3 lo = F3((o & 1u) ? pn.mid.x : pn.lo.x, (o & 2u) ? pn.mid.y : pn.lo.y, (o & 4u) ? pn.mid.z : pn.lo.z);
                const f3 hi = F3((o & 1u) ? pn.hi.x : pn.mid.x, (o & 2u) ? pn.hi.y : pn.mid.y, (o & 4u) ? pn.hi.z : pn.mid.z);
                const f3 t1 = (lo - r.pos) * r.inv, t2 = (hi - r.pos) * r.inv;
                const float tmin = fmaxf(fmaxf(fminf(t1.x, t2.x), fminf(t1.y, t2.y)), fminf(t1.z, t2.z));
                const float tmax = fminf(fminf(fmaxf(t1.x, t2.x), fmaxf(t1.y, t2.y)), fmaxf(t1.z, t2.z));
                const bool through = tmax > 0.0f && tmin < tmax;
                uncertain = self ? !(through && tmin <= closest) : through;
            }
            if (COUNT) cn.cand_checks++;
            if (uncertain) {                     // step 3
                if (COUNT) cn.fallbacks++;
                x.post_q = x.post_end = 0; x.nf = false; x.tie = false;
                closest = x.closest0; closest_tri = 0; tri_hit = false;
                w.begin(st, 0u);
            } else if (++x.post_q == x.post_end) {
                x.post_q = x.post_end = 0;
                active = false;
            }
            return;
        }
        const bool nf_pass = EXACT && x.nf;
        if (w.node.child_mask != 0) {
            uint32_t pass = child_pass_octants<EXACT != 0, !COUNT>(r, w.node.lo, w.node.mid, w.node.hi, w.node.child_mask, closest, nf_pass ? x.band : 0.0f, &x.tie);
            // (the permutation as a 2 KB table in LDS -- one byte load instead of fifteen integer instructions -- measured 1.5 % faster on the
            // 100 k-primitive frame and 5.6 % SLOWER on the 1.28 M-triangle one: the load sits in every visit's chain)
            if (near_first) pass = permute_octants(pass, w.key);
            if (pass) st.push_level(w.node.first_child, pass);
            w.node.child_mask = 0;
        }
        if (st.empty()) {
            if (nf_pass && tri_hit) {
                const bool redo = x.tie;
                if (!redo && EXACT == 2 && x.slot) {
                    x.post_q = S.tris[closest_tri].dup_first; x.post_end = S.tris[closest_tri].dup_pos + 1u;   // (at least the kept reference itself)
                    return;
                }
                if (redo) {
                    if (COUNT) cn.fallbacks++;
                    x.nf = false; x.tie = false;
                    closest = x.closest0; closest_tri = 0; tri_hit = false;
                    w.begin(st, 0u);
                    return;
                }
            }
            active = false;
        } else {
            w.node = load_node<COUNT>(S, top, st.pop_node(w.key), cn);
            if (EXACT) asm volatile("" : "+v"(w.node.first_id));    // (fetched with the rest of the record, not in a round trip of its own where a leaf needs it)
            if (COUNT) { cn.pops++; cn.tri_tests += w.node.tri_count; cn.box_tests += __popc(w.node.child_mask); }
            w.tri_i = w.node.tri_start; w.tri_end = w.node.tri_start + w.node.tri_count;
            if (w.node.child_mask == 0 && w.node.tri_count != 0) {   // a leaf's record holds its first triangle
                accept_hit<EXACT>(ray_triangle(r, inline_triangle(w.node)), w.tri_i, w.node.first_id, w.node.rank, closest, closest_tri, tri_hit, x);
                w.tri_i++;
            }
        }
    }
}

DEVI float sign1(float y) { return y > 0.0f ? 1.0f : (y < 0.0f ? -1.0f : 0.0f); }

// frag:284-318 with the constant arguments of frag:518; the compiler the reference runs under pulls the folded
// constant ra*ra*caca inward: c = (caca*dot(oc,oc) - K) - caoc*caoc.
DEVI float ray_cylinder(const Ray &r, float closest, f3 &normal) {
    const f3 pa = F3(2.1f, 0.1f, -2.0f), pb = F3(1.9f, 0.5f, -1.9f);
    const float ra = 0.08f;
    const f3 ca = pb - pa;
    const f3 oc = r.pos - pa;
    const float caca = dot3(ca, ca);
    const float card = dot3(ca, r.dir);
    const float caoc = dot3(ca, oc);
    const float a = caca - card * card;
    const float b = caca * dot3(oc, r.dir) - caoc * card;
    const float c = (caca * dot3(oc, oc) - ra * ra * caca) - caoc * caoc;
    float h = b * b - a * c;
    if (h < 0.0f) return __builtin_inff();
    h = sqrt_ieee(h);
    float d = div_ieee(-b - h, a);
    const float y = caoc + d * card;
    if (y > 0.0f && y < caca && d >= 0.0f && d <= closest) {
        const f3 p = (oc + r.dir * d) - F3(div_ieee(ca.x * y, caca), div_ieee(ca.y * y, caca), div_ieee(ca.z * y, caca));
        normal = F3(div_ieee(p.x, ra), div_ieee(p.y, ra), div_ieee(p.z, ra));
        return d;
    }
    d = div_ieee((y < 0.0f ? 0.0f : caca) - caoc, card);
    if (__builtin_fabsf(b + a * d) < h && d >= 0.0f && d <= closest) {
        const float sy = sign1(y);
        normal = normalize3(F3(div_ieee(ca.x * sy, caca), div_ieee(ca.y * sy, caca), div_ieee(ca.z * sy, caca)));
        return d;
    }
    return __builtin_inff();
}

// The part of RayPortal that does not depend on the ray (frag:331-337): the rectangle's in-plane axes.  Computed once
// per workgroup with the same operations every fragment would repeat, and kept in LDS.
struct PortalBasis { float rx, ry, rz, ux, uy, uz; uint32_t ok, pad; };
DEVI PortalBasis portal_basis(f3 pn) {
    PortalBasis b;
    f3 right = cross3(pn, F3(0.0f, 1.0f, 0.0f));
    b.ok = length3(right) <= 0.0001f ? 0u : 1u;
    right = normalize3(right);
    const f3 up = normalize3(cross3(right, pn));
    b.rx = right.x; b.ry = right.y; b.rz = right.z; b.ux = up.x; b.uy = up.y; b.uz = up.z; b.pad = 0u;
    return b;
}
DEVI void stage_portals(PortalBasis *lds2, const rt_uniforms &U) {
    if (threadIdx.x >= blockDim.x - 2u) {   // the last two threads: the first ones are busy copying the sphere data
        const uint32_t which = threadIdx.x - (blockDim.x - 2u);
        const float *d = which == 0 ? U.portal_direction_1 : U.portal_direction_2;
        lds2[which] = portal_basis(F3(d[0], d[1], d[2]));
    }
    __syncthreads();
}
// frag:320-350
DEVI float ray_portal(const Ray &r, f3 ppos, f3 pn, const PortalBasis &pb, float pw, float ph, float closest) {
    const float d = dot3(pn, r.dir);
    if (__builtin_fabsf(d) <= 0.0001f) return __builtin_inff();
    const float num = dot3(ppos - r.pos, pn);
    // The plane behind the ray: t = num / d is negative -- and a normal number, not flushed to -0 -- whenever the signs
    // differ and |num| >= |d| * 2^-100, so the quotient need not be formed to know that `t < 0` rejects (NaNs fall through).
    if ((num < 0.0f) != (d < 0.0f) && __builtin_fabsf(num) >= __builtin_fabsf(d) * 0x1p-100f) return __builtin_inff();
    const float t = div_ieee(num, d);
    if (t < 0.0f || t > closest) return __builtin_inff();
    const f3 ip = r.pos + r.dir * t;
    if (!pb.ok) return __builtin_inff();
    const f3 c = ip - ppos;
    if (__builtin_fabsf(dot3(F3(pb.rx, pb.ry, pb.rz), c)) < 0.5f * pw && __builtin_fabsf(dot3(F3(pb.ux, pb.uy, pb.uz), c)) < 0.5f * ph) return t;
    return __builtin_inff();
}

// frag:215-230 with the two edges v0 = b - a and v1 = c - a taken from the shading record (RtShadeRec: the same subtractions, done once per scene)
DEVI f3 barycentric(f3 p, f3 a, f3 v0, f3 v1) {
    const f3 v2 = p - a;
    const float d00 = dot3(v0, v0), d01 = dot3(v0, v1), d11 = dot3(v1, v1), d20 = dot3(v2, v0), d21 = dot3(v2, v1);
    const float denom = d00 * d11 - d01 * d01;
    const float v = div_ieee(d11 * d20 - d01 * d21, denom);
    const float w = div_ieee(d00 * d21 - d01 * d20, denom);
    const float u = 1.0f - v - w;
    return F3(u, v, w);
}

// Everything of FindIntersection after the octree loop (frag:517-562): cylinder, portals, hit assembly.
template <bool COUNT>
DEVI Hit finish_intersection(const RtDeviceScene &S, const rt_uniforms &U, const PortalBasis *portals, const Ray &r, float closest, uint32_t closest_i, uint32_t closest_tri,
                             bool tri_hit, Counters &cn) {
    bool cyl_hit = false;
    f3 cyl_n = F3(0.0f, 0.0f, 0.0f);
    const float cyl_t = ray_cylinder(r, closest, cyl_n);
    if (!is_inf(cyl_t)) { cyl_hit = true; closest = cyl_t; }

    const f3 pp1 = F3(U.portal_position_1[0], U.portal_position_1[1], U.portal_position_1[2]);
    const f3 pd1 = F3(U.portal_direction_1[0], U.portal_direction_1[1], U.portal_direction_1[2]);
    const f3 pp2 = F3(U.portal_position_2[0], U.portal_position_2[1], U.portal_position_2[2]);
    const f3 pd2 = F3(U.portal_direction_2[0], U.portal_direction_2[1], U.portal_direction_2[2]);
    const uint32_t first = opaque(0u);   // keeps the loads of the two bases next to their use
    const float p1 = ray_portal(r, pp1, pd1, portals[first], U.portal_width, U.portal_height, closest);
    const float p2 = ray_portal(r, pp2, pd2, portals[first + 1u], U.portal_width, U.portal_height, closest);

    Hit h;
    h.has_hit = false; h.position = F3(0, 0, 0); h.normal = F3(0, 0, 0); h.material_id = 0; h.portal_id = 0; h.kind = RT_HIT_MISS; h.slot = 0; h.t = closest;
    if (!is_inf(p1) && p1 < p2) {
        h.has_hit = true; h.position = r.pos + r.dir * p1; h.normal = pd1; h.portal_id = 1;
        h.kind = RT_HIT_PORTAL_1; h.t = p1;
        return h;
    }
    if (!is_inf(p2) && p2 < p1) {
        h.has_hit = true; h.position = r.pos + r.dir * p2; h.normal = pd2; h.portal_id = 2;
        h.kind = RT_HIT_PORTAL_2; h.t = p2;
        return h;
    }
    if (is_inf(closest)) {
        return h;
    }
    h.has_hit = true;
    h.position = r.pos + r.dir * closest;
    if (cyl_hit) {
        h.normal = cyl_n;
        h.kind = RT_HIT_CYLINDER;
    } else if (tri_hit) {
        const float4 *sr = reinterpret_cast<const float4 *>(S.shade + closest_tri);      // frag:541-549 from one record
        const float4 q0 = sr[0], q1 = sr[1], q2 = sr[2], q3 = sr[3], q4 = sr[4];
        const f3 uvw = barycentric(h.position, F3(q0.x, q0.y, q0.z), F3(q0.w, q1.x, q1.y), F3(q1.z, q1.w, q2.x));
        h.normal = (F3(q2.y, q2.z, q2.w) * uvw.x + F3(q3.x, q3.y, q3.z) * uvw.y) + F3(q3.w, q4.x, q4.y) * uvw.z;
        h.material_id = f2u(q4.z);
        if (COUNT) cn.tri_shaded++;
        h.kind = RT_HIT_TRIANGLE; h.slot = closest_tri;
    } else {
        const float4 s = c_spheres[closest_i];
        h.normal = normalize3(h.position - F3(s.x, s.y, s.z));
        h.material_id = closest_i % 7u;
        h.kind = RT_HIT_SPHERE; h.slot = closest_i;
    }
    return h;
}

// ------------------------------------------------------------------------------------------------
// environment lookup (frag:662): RGBA8 cubemap, LINEAR, CLAMP_TO_EDGE, non-seamless, filtered in 8-bit fixed
// point.  Major-axis selection uses UNORDERED comparisons and sign BITS (NaN directions from total internal
// reflection reach this lookup: frag:639-650); on x86 an invalid operation yields a NEGATIVE quiet NaN, which
// this hardware does not, so NaN components are given that sign here.
// ------------------------------------------------------------------------------------------------
DEVI int cvt_rne_x86(float f) {
    if (!(__builtin_fabsf(f) < 2147483648.0f)) return (int)0x80000000;
    return (int)__builtin_rintf(f);
}
DEVI float x86_nan(float v) { return v != v ? u2f(0xffc00000u) : v; }
DEVI f3 sample_sky(const RtDeviceScene &S, f3 d) {
    if (!S.sky) return F3(0.0f, 0.0f, 0.0f);
    d = F3(x86_nan(d.x), x86_nan(d.y), x86_nan(d.z));
    const float ax = __builtin_fabsf(d.x), ay = __builtin_fabsf(d.y), az = __builtin_fabsf(d.z);
    const bool x_over_y = !(ax <= ay);
    const float maxxy = ax > ay ? ax : ay;
    const bool z_sel = !(az < maxxy);
    int face; float sc, tc, ma;
    if (z_sel) { ma = d.z; const bool neg = (f2u(ma) >> 31) != 0; face = 4 + (neg ? 1 : 0); sc = neg ? -d.x : d.x; tc = -d.y; }
    else if (x_over_y) { ma = d.x; const bool neg = (f2u(ma) >> 31) != 0; face = 0 + (neg ? 1 : 0); sc = neg ? d.z : -d.z; tc = -d.y; }
    else { ma = d.y; const bool neg = (f2u(ma) >> 31) != 0; face = 2 + (neg ? 1 : 0); sc = d.x; tc = neg ? -d.z : d.z; }
    // projection: coord * (0.5 / |ma|) + 0.5; 0.5 / x and 0.5 * (1 / x) round alike (a power of two scales exactly, and a
    // result too small to be normal is flushed either way)
    const float ima = 0.5f * rcp_ieee(__builtin_fabsf(ma));
    const float s = ima * sc + 0.5f;
    const float t = ima * tc + 0.5f;
    const int W = S.sky_w, H = S.sky_h;
    const int Ui = (int)((uint32_t)cvt_rne_x86(s * (float)W * 256.0f) - 128u);
    const int Vi = (int)((uint32_t)cvt_rne_x86(t * (float)H * 256.0f) - 128u);
    int i0 = Ui >> 8, j0 = Vi >> 8;
    const int wu = Ui & 255, wv = Vi & 255;
    int i1 = i0 + 1, j1 = j0 + 1;
    i0 = min(max(i0, 0), W - 1); i1 = min(max(i1, 0), W - 1);
    j0 = min(max(j0, 0), H - 1); j1 = min(max(j1, 0), H - 1);
    const uchar4 *f = reinterpret_cast<const uchar4 *>(S.sky) + (size_t)face * W * H;
    const uchar4 p00 = f[(size_t)j0 * W + i0], p10 = f[(size_t)j0 * W + i1], p01 = f[(size_t)j1 * W + i0], p11 = f[(size_t)j1 * W + i1];
    auto lerp8 = [](int a, int b, int w) { return (a * (256 - w) + b * w + 128) >> 8; };
    const int rr = lerp8(lerp8(p00.x, p10.x, wu), lerp8(p01.x, p11.x, wu), wv);
    const int gg = lerp8(lerp8(p00.y, p10.y, wu), lerp8(p01.y, p11.y, wu), wv);
    const int bb = lerp8(lerp8(p00.z, p10.z, wu), lerp8(p01.z, p11.z, wu), wv);
    const float k = 1.0f / 255.0f;   // unorm8 -> float multiplies by the rounded reciprocal
    return F3((float)rr * k, (float)gg * k, (float)bb * k);
}

// ------------------------------------------------------------------------------------------------
// ray generation, scattering, tonemap (frag:566-687)
// ------------------------------------------------------------------------------------------------
DEVI f3 mat_dir(const float *M, f3 v) {     // (M * vec4(v, 0)).xyz
    return F3((M[0] * v.x + M[4] * v.y) + M[8] * v.z, (M[1] * v.x + M[5] * v.y) + M[9] * v.z, (M[2] * v.x + M[6] * v.y) + M[10] * v.z);
}
DEVI f3 mat_point(const float *M, f3 v) {   // (M * vec4(v, 1)).xyz
    return F3(((M[0] * v.x + M[4] * v.y) + M[8] * v.z) + M[12], ((M[1] * v.x + M[5] * v.y) + M[9] * v.z) + M[13], ((M[2] * v.x + M[6] * v.y) + M[10] * v.z) + M[14]);
}
DEVI float nonlinear_depth(float d, float zn, float zf) { return div_ieee(zn * zf - d * zf, d * (zn - zf)); }   // frag:352-354
DEVI float clamp01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

// frag:674-684.  Returns the primary ray direction and the re-based seed handed to the path loop.
DEVI void generate_ray(const rt_uniforms &U, float ndc_x, float ndc_y, f3 &dir, float &seed_base) {
    const float *M = U.inv_view_proj_mat;
    const f3 cam = F3(U.camera_position[0], U.camera_position[1], U.camera_position[2]);
    float p[4];
#pragma unroll
    for (int r = 0; r < 4; r++) p[r] = ((M[r] * ndc_x + (-M[8 + r])) + M[4 + r] * ndc_y) + M[12 + r];   // frag:676 as evaluated
    const f3 pp = F3(div_ieee(p[0], p[3]), div_ieee(p[1], p[3]), div_ieee(p[2], p[3]));
    const float s1 = (float)base_hash(f2u(pp.x - U.time), f2u(pp.y - U.time)) / 4294967296.0f;   // frag:679
    Seed seed;
    seed.base = (float)base_hash(f2u(s1), f2u(s1)) / 4294967296.0f;                               // frag:680
    seed.k = 0;
    const f3 jit = random_in_unit_sphere(seed);
    dir = normalize3(pp - (cam + jit * U.blur_amount));                                           // frag:682
    seed.rebase();
    seed_base = seed.base;
}

// First-segment depth (frag:575-581)
DEVI float first_hit_depth(const rt_uniforms &U, const Hit &h, const Ray &ray) {
    if (h.has_hit) return clamp01(nonlinear_depth(length3(h.position - ray.pos), U.z_near, U.z_far));
    return clamp01(nonlinear_depth(U.z_far, U.z_near, U.z_far));
}

// One iteration of the path loop after the intersection (frag:583-659) for a hit: updates ray, colour and seed.
// `last`: this was the path's final segment (frag:571: the loop ends), so the outgoing ray and the seed are never used
// again and only the colour is updated -- the same colour: which factor applies depends on one random number drawn before
// the direction is sampled.
DEVI void scatter(const rt_uniforms &U, const Hit &h, Ray &ray, f3 &color, Seed &seed, bool last) {
    if (h.portal_id != 0) {
        const bool one = h.portal_id == 1;
        const f3 pp1 = F3(U.portal_position_1[0], U.portal_position_1[1], U.portal_position_1[2]);
        const f3 pd1 = F3(U.portal_direction_1[0], U.portal_direction_1[1], U.portal_direction_1[2]);
        const f3 pp2 = F3(U.portal_position_2[0], U.portal_position_2[1], U.portal_position_2[2]);
        const f3 pd2 = F3(U.portal_direction_2[0], U.portal_direction_2[1], U.portal_direction_2[2]);
        const f3 pd = one ? pd1 : pd2, psrc = one ? pp1 : pp2, pdst = one ? pp2 : pp1;
        const float *PM = one ? U.portal_1_to_2 : U.portal_2_to_1;
        color = color * (dot3(ray.dir, pd) < 0.0f ? 0.5f : 0.05f);
        if (last) return;
        ray.pos = mat_point(PM, h.position - psrc) + pdst;
        ray.dir = normalize3(mat_dir(PM, ray.dir));
        ray.pos = ray.pos + ray.dir * 0.001f;
    } else {
        const Material m = c_materials[h.material_id];
        const f3 mcol = F3(m.r, m.g, m.b);
        // The three materials are written as one sequence instead of three separate branches: the random vector of the rough
        // reflection / refraction (a cube root, a sine and a cosine) is drawn by metals, dielectrics and the specular case of the
        // diffuse material alike, and lanes of different materials share one copy of that code instead of running three copies one
        // after the other.  Per lane the operations and the order of the random draws are those of frag:604-657.
        bool diffuse = false;
        if (m.type == LAMBERTIAN) {
            const float F = fresnel_schlick_roughness(fmaxf(-dot3(ray.dir, h.normal), 0.0f), 0.04f, m.roughness);
            diffuse = hash1(seed) > F;
            if (diffuse) color = color * mcol;
        } else if (m.type == METAL) {
            color = color * mcol;
        }
        if (last) return;       // (a dielectric leaves the colour alone)
        f3 rough = F3(0.0f, 0.0f, 0.0f);
        if (!diffuse) rough = random_in_unit_sphere(seed) * m.roughness;
        if (m.type == DIELECTRIC) {
            float ri, cosine; f3 on;
            const float dn = dot3(ray.dir, h.normal);
            if (dn > 0.0f) {
                ri = m.refractive_index;
                cosine = sqrt_ieee(1.0f - ri * ri * (1.0f - dn * dn));
                on = -h.normal;
            } else {
                ri = rcp_ieee(m.refractive_index);
                cosine = -dn;
                on = h.normal;
            }
            const f3 md = ray.dir + rough;
            const f3 refr = normalize3(refract3(md, on, ri));
            if (refr.x != 0.0f || refr.y != 0.0f || refr.z != 0.0f) {   // true for NaN (total internal reflection)
                const float rr = div_ieee(1.0f - ri, 1.0f + ri);
                const float F = fresnel_schlick_roughness(cosine, rr * rr, m.roughness);
                if (hash1(seed) > F) { ray.pos = h.position - on * 0.001f; ray.dir = refr; }
                else { ray.pos = h.position + on * 0.001f; ray.dir = normalize3(reflect3(md, on)); }
            } else {
                ray.pos = h.position - on * 0.001f;
                ray.dir = normalize3(reflect3(md, on));
            }
        } else {
            ray.pos = h.position + h.normal * 0.001f;
            if (diffuse) ray.dir = random_cos_weighted_hemisphere_direction(h.normal, seed);
            else ray.dir = normalize3(reflect3(ray.dir, h.normal) + rough);
        }
    }
    seed.rebase();
}

// frag:667-668
DEVI f3 tonemap(f3 color) {
    float c[3] = {color.x, color.y, color.z};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float x = fmaxf(0.0f, c[k] - 0.004f);
        c[k] = div_ieee(x * (6.2f * x + 0.5f), x * (6.2f * x + 1.7f) + 0.06f);
    }
    return F3(c[0], c[1], c[2]);
}

// Colour attachment write with the blend state of App.cpp:49,104: dst = src*(1/n) + dst*(1 - 1/n) on all four
// channels (src alpha = 0), or a plain store when blending is off.
DEVI float4 blend_sample(float4 dst, f3 c, uint32_t accum_index, uint32_t blend) {
    float4 out;
    if (blend) {
        const float a = rcp_ieee((float)accum_index), ia = 1.0f - a;
        out.x = c.x * a + dst.x * ia; out.y = c.y * a + dst.y * ia; out.z = c.z * a + dst.z * ia; out.w = 0.0f * a + dst.w * ia;
    } else {
        out.x = c.x; out.y = c.y; out.z = c.z; out.w = 0.0f;
    }
    return out;
}
DEVI void write_pixel(const RtLaunch &L, size_t pi, f3 c) {
    float4 *dstp = reinterpret_cast<float4 *>(L.color) + pi;
    const float4 dst = L.blend ? *dstp : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    *dstp = blend_sample(dst, c, L.accum_index, L.blend);
}

// Pixel addressing shared by both pipelines: one wave per 8x8 pixel tile, 4 waves (a 16x16 tile) per workgroup;
// workgroups are renumbered so that the 8 round-robin dispatch classes (blockIdx % 8: the workgroups that share an
// XCD's L2) each own a contiguous range of tiles.
struct PixelId { uint32_t x, y, lr; bool active; size_t fi, pi; };
// Pixel `tid` (0..255: four 8x8 quarters of 64) of the 16x16 tile at (tile_x, tile_y) of this launch's local rows.
DEVI PixelId pixel_of_tile(const RtLaunch &L, uint32_t tile_x, uint32_t tile_y, uint32_t tid) {
    const uint32_t wave = tid >> 6, lane = tid & 63u;
    PixelId p;
    p.x = tile_x * 16u + (wave & 1u) * 8u + (lane & 7u);
    p.lr = tile_y * 16u + (wave >> 1) * 8u + (lane >> 3);
    // local row -> frame row: strip j of this part is frame strip j*parts + part
    p.y = ((p.lr / L.band_rows) * L.band_parts + L.band_part) * L.band_rows + (p.lr % L.band_rows);
    p.active = p.x < L.width && p.lr < L.n_local_rows && p.y < L.height;
    p.fi = (size_t)p.y * L.width + p.x;                                   // full-frame pixel index (NDC tables)
    // index into the target planes; in compact mode the rows of this launch's strips are interleaved with those of
    // the other sub-frames of the same part (sub-frame j of K owns every K-th strip of the part)
    const uint32_t part_lr = ((p.lr / L.band_rows) * L.sub_parts + L.sub_part) * L.band_rows + (p.lr % L.band_rows);
    p.pi = L.compact_rows ? (size_t)part_lr * L.width + p.x : p.fi;
    return p;
}
// Workgroups are dealt to the 8 XCDs round-robin by their index; this renumbering (a bijection on [0, n)) gives each
// XCD (index % 8) a contiguous range, so that neighbouring pieces of the frame share an L2.
DEVI uint32_t xcd_contiguous(uint32_t raw, uint32_t n) {
    const uint32_t q = n / 8u, rem = n % 8u, cls = raw % 8u, idx = raw / 8u;
    return (cls < rem ? cls * (q + 1u) : rem * (q + 1u) + (cls - rem) * q) + idx;
}
// Tile order of the one-lane-per-pixel and the persistent path kernel: the tiles of the rectangle L.hot (where the octree projects to: the
// pixels whose paths can be long) come first, then the rows below it, the rows above it, and the tiles left and right
// of it, so that the expensive paths start early and the frame ends on cheap ones.  k -> (tile_x, tile_y), bijective.
DEVI void path_tile(const RtLaunch &L, uint32_t k, uint32_t &tile_x, uint32_t &tile_y) {
    const uint32_t tiles_x = (L.width + 15u) / 16u, tiles_y = (L.n_local_rows + 15u) / 16u;
    const uint32_t tx0 = L.hot[0], ty0 = L.hot[1], tx1 = min(L.hot[2], tiles_x - 1u), ty1 = min(L.hot[3], tiles_y - 1u);
    if (tx1 < tx0 || ty1 < ty0) { tile_x = k % tiles_x; tile_y = k / tiles_x; return; }
    const uint32_t w = tx1 - tx0 + 1u, h = ty1 - ty0 + 1u;
    if (k < w * h) { tile_x = tx0 + k % w; tile_y = ty0 + k / w; return; }
    k -= w * h;
    if (k < ty0 * tiles_x) { tile_x = k % tiles_x; tile_y = k / tiles_x; return; }
    k -= ty0 * tiles_x;
    const uint32_t above = (tiles_y - 1u - ty1) * tiles_x;
    if (k < above) { tile_x = k % tiles_x; tile_y = ty1 + 1u + k / tiles_x; return; }
    k -= above;
    const uint32_t side = tiles_x - w;   // > 0 here: otherwise the three cases above cover every tile
    const uint32_t c = k % side;
    tile_x = c < tx0 ? c : c + w;
    tile_y = ty0 + k / side;
}
DEVI void pixel_ndc(const RtLaunch &L, const PixelId &p, float &nx, float &ny) {
    nx = L.ndc_x ? L.ndc_x[p.fi] : div_ieee(2.0f * ((float)p.x + 0.5f), (float)L.width) - 1.0f;
    ny = L.ndc_y ? L.ndc_y[p.fi] : div_ieee(2.0f * ((float)p.y + 0.5f), (float)L.height) - 1.0f;
}

DEVI uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
DEVI void flush_counters(unsigned long long *ctr, const Counters &cn) {
    const uint32_t v[10] = {wave_sum(cn.rays), wave_sum(cn.pops), wave_sum(cn.tri_tests), wave_sum(cn.box_tests), wave_sum(cn.tri_shaded), wave_sum(cn.entered), wave_sum(cn.sph_tests),
                            wave_sum(cn.lds_pops), wave_sum(cn.fallbacks), wave_sum(cn.cand_checks)};
    if ((threadIdx.x & 63u) == 0) {
#pragma unroll
        for (int k = 0; k < 10; k++) if (v[k]) atomicAdd(&ctr[k], (unsigned long long)v[k]);
    }
}

#ifdef RT_FULL_SPHERE_LOOP   // A/B switch: test all 84 spheres for every ray
#define SPHERE_LOOP(PRIMARY, lds, ray, closest, closest_i, n_tests) do { if (n_tests) *(n_tests) += NUM_SPHERES; if (PRIMARY) intersect_spheres_primary((lds).tab, (ray).dir, closest, closest_i); else intersect_spheres((lds).tab, ray, closest, closest_i); } while (0)
#else
#define SPHERE_LOOP(PRIMARY, lds, ray, closest, closest_i, n_tests) intersect_spheres_culled<PRIMARY>(&(lds), ray, closest, closest_i, n_tests)
#endif

// ================================================================================================
// Pipeline 1: one kernel, one lane per pixel for the whole path (kept as the simple cross-check).
// ================================================================================================
// Resident waves per SIMD the register allocator must make room for.  The kernel hides the latency of its dependent
// node and triangle loads only through other waves: 4 / 5 / 6 / 7 waves per SIMD give 0.74 / 0.65 / 0.59 / 0.57 ms per 4K
// frame.  Seven is what a CU's LDS allows when the stack has at most 19 levels (76 bytes per lane, plus 3.4 KB of sphere
// and portal data shared by the four waves of a workgroup: seven 256-thread workgroups); it leaves 72 registers, one of
// which is spilled.  A 20-level stack fits six workgroups; the variant for very deep trees needs 46 KB per workgroup,
// three fit, and it keeps its registers.
#ifndef RENDER_WAVES_PER_SIMD
#define RENDER_WAVES_PER_SIMD 6
#endif
#define RENDER_OCCUPANCY __attribute__((amdgpu_waves_per_eu(LEVELS <= 19 ? 7 : LEVELS <= 20 ? RENDER_WAVES_PER_SIMD : 3, LEVELS <= 19 ? 7 : LEVELS <= 20 ? RENDER_WAVES_PER_SIMD : 3)))
#define RENDER_BLOCK 256   // threads per workgroup = pixels per unit of the cost map (a 16x16 tile)
// BATCH: the launch traces B->n consecutive frames (see RtBatch), the expensive units of all frames still first.
template <int LEVELS, bool COUNT, int BLOCK, bool BATCH>
DEVI void render_body(const RtDeviceScene &S, const RtLaunch &L, const RtBatch *B) {
    __shared__ uint32_t lds_stack[LEVELS * BLOCK];
    __shared__ SphereLds lds_sph;
    constexpr uint32_t WAVES = BLOCK / 64u;
    __shared__ PortalBasis lds_portals_all[BATCH ? 2u * WAVES : 2u];
    // BATCH: the waves of a workgroup render the SAME 8x8 pixels for different frames of the batch.  Their primary rays
    // differ only by the sub-pixel jitter, so they walk the same nodes and triangles at the same time and share the
    // CU's vector cache.  Workgroup b: frames (b % groups) * WAVES + wave, 8x8 tile b / groups of the cost-ordered list
    // (a unit of that list is UNIT_TILES such tiles, whatever the workgroup size).  When one workgroup holds all frames
    // of the batch (groups == 1) it also blends them into the target, in frame order, through LDS; otherwise the samples
    // are staged for blend_batch_kernel.
    constexpr uint32_t UNIT_TILES = RENDER_BLOCK / 64u;
    const uint32_t groups = BATCH ? (B->n + WAVES - 1u) / WAVES : 1u;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t frame_raw = BATCH ? (blockIdx.x % groups) * WAVES + wave : 0u;
    const bool frame_ok = !BATCH || frame_raw < B->n;
    const uint32_t frame = BATCH ? (uint32_t)__builtin_amdgcn_readfirstlane((int)min(frame_raw, B->n - 1u)) : 0u;
    const rt_uniforms &U = BATCH ? B->u[frame] : L.u;
    stage_spheres(&lds_sph, c_spheres);
    if (BATCH) {
        if (threadIdx.x >= BLOCK - 2u * WAVES) {
            const uint32_t which = threadIdx.x - (BLOCK - 2u * WAVES);
            const uint32_t f = min((blockIdx.x % groups) * WAVES + which / 2u, B->n - 1u);
            const float *d = (which & 1u) == 0 ? B->u[f].portal_direction_1 : B->u[f].portal_direction_2;
            lds_portals_all[which] = portal_basis(F3(d[0], d[1], d[2]));
        }
        __syncthreads();
    } else {
        stage_portals(lds_portals_all, U);
    }
    const PortalBasis *lds_portals = lds_portals_all + (BATCH ? 2u * wave : 0u);
    LevelStackT<BLOCK> st;
    st.col = &lds_stack[threadIdx.x];
    st.sp = 0;
    // workgroups start in index order: the tiles where the octree projects to come first (path_tile), so the expensive
    // paths are under way early and the kernel does not end on them (measured: 1.04 -> 0.95 ms per 4K frame)
    // With a cost order from the previous frame of the same view (L.order) the most expensive units start first.
    const unsigned long long t_start = L.cost ? wall_clock64() : 0ull;
    const uint32_t q = BATCH ? blockIdx.x / groups : 0u;                       // BATCH: 8x8 tile index in the ordered list
    const uint32_t slot = BATCH ? q / UNIT_TILES : blockIdx.x;
    // The order lists 16x16 tiles by their position in the frame (tile_y * tiles_x + tile_x), and so does the cost map: a
    // tile keeps its cost when the rectangle the octree projects to -- which only decides the start order of a geometry's
    // first launch -- moves with the camera.
    const uint32_t tiles_x = (L.width + 15u) / 16u;
    uint32_t tile_x, tile_y;
    if (L.order) { const uint32_t id = L.order[slot]; tile_x = id % tiles_x; tile_y = id / tiles_x; }
    else path_tile(L, slot, tile_x, tile_y);
    const uint32_t unit = tile_y * tiles_x + tile_x;
    PixelId p = pixel_of_tile(L, tile_x, tile_y, BATCH ? (q % UNIT_TILES) * 64u + lane : threadIdx.x);
    const bool in_frame = p.active;
    p.active = p.active && frame_ok;
    const bool fused_blend = BATCH && groups == 1u;
    f3 out = F3(0.0f, 0.0f, 0.0f);
    Counters cn = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (p.active) {
        float nx, ny;
        pixel_ndc(L, p, nx, ny);
        Ray ray;
        Seed seed;
        seed.k = 0;
        ray.pos = F3(U.camera_position[0], U.camera_position[1], U.camera_position[2]);
        generate_ray(U, nx, ny, ray.dir, seed.base);
        ray.inv = rcp3(ray.dir);
        f3 color = F3(1.0f, 1.0f, 1.0f);
        uint32_t nrays = 0;
        if (U.max_recursion_limit == 0 && (!BATCH || frame + 1u == B->n)) {   // no segment at all: the planes of the first segment keep their initial values
            if (L.depth) L.depth[p.pi] = 1.0f;
            if (L.hit_kind) L.hit_kind[p.pi] = RT_HIT_MISS;
            if (L.hit_slot) L.hit_slot[p.pi] = 0;
            if (L.hit_t) L.hit_t[p.pi] = __builtin_inff();
        }
        for (uint32_t i = 0; i < U.max_recursion_limit; i++) {
            float closest = __builtin_inff();
            uint32_t closest_i = 0, closest_tri = 0;
            bool tri_hit = false;
            if (COUNT) cn.rays++;
            SPHERE_LOOP(false, lds_sph, ray, closest, closest_i, COUNT ? &cn.sph_tests : nullptr);
            if (root_box_test(S, ray, closest)) {
                if (COUNT) cn.entered++;
                // (a lane per visit here: with a wave's rays this coherent the wave-wide walk_step measured 10 % slower)
                const bool near_first = L.near_first == 1u;   // (the exact near-first mode is built into the wave-wide traversal only: here it runs the reference's order, which gives the same result)
                const uint32_t key = order_key(ray, near_first);
                st.reset_to_root(key);
                while (!st.empty()) visit_node<COUNT>(S, ray, st, key, near_first, closest, closest_tri, tri_hit, cn);
            }
            const Hit h = finish_intersection<COUNT>(S, U, lds_portals, ray, closest, closest_i, closest_tri, tri_hit, cn);
            nrays++;
            if (i == 0) {   // the first segment's planes are stored at once instead of being carried through the path
                const uint32_t pix = opaque((uint32_t)p.pi);   // (addresses formed here, not ahead of the loop)
                if (BATCH && frame + 1u != B->n) { /* the planes keep the last frame's values */ } else {
                if (L.depth) L.depth[pix] = first_hit_depth(U, h, ray);
                if (L.hit_kind) L.hit_kind[pix] = h.kind;
                if (L.hit_slot) L.hit_slot[pix] = h.slot;
                if (L.hit_t) L.hit_t[pix] = h.t;
                }
            }
            if (!h.has_hit) { color = color * sample_sky(S, ray.dir); break; }
            const bool last = i + 1u == U.max_recursion_limit;
            scatter(U, h, ray, color, seed, last);
            if (!last) ray.inv = rcp3(ray.dir);
        }
        out = tonemap(color);
        if (!BATCH) write_pixel(L, p.pi, out);
        else if (!fused_blend) B->stage[(size_t)frame * B->stage_stride + (size_t)p.lr * L.width + p.x] = make_float4(out.x, out.y, out.z, 0.0f);
        if (L.n_rays && (!BATCH || frame + 1u == B->n)) L.n_rays[p.pi] = nrays;
    }
    if (L.cost && (threadIdx.x & 63u) == 0) atomicMax(&L.cost[unit], (uint32_t)min(wall_clock64() - t_start, 0xFFFFFFFEull) + 1u);
    if (fused_blend) {
        // the traversal stacks are dead once every wave is here: their LDS carries the samples to the first wave, which
        // applies them to its 64 pixels frame after frame (App.cpp:49,102-105)
        __syncthreads();
        f3 *samples = reinterpret_cast<f3 *>(lds_stack);
        samples[threadIdx.x] = out;
        __syncthreads();
        if (wave == 0 && in_frame) {
            float4 *dstp = reinterpret_cast<float4 *>(L.color) + p.pi;
            float4 dst = *dstp;
            for (uint32_t f = 0; f < B->n; f++) dst = blend_sample(dst, samples[f * 64u + lane], B->accum_index[f], B->blend[f]);
            *dstp = dst;
        }
    }
    if (COUNT && L.counters) flush_counters(L.counters, cn);
}

template <int LEVELS, bool COUNT, int BLOCK>
__global__ __launch_bounds__(BLOCK) RENDER_OCCUPANCY void render_kernel(RtDeviceScene S, RtLaunch L) {
    render_body<LEVELS, COUNT, BLOCK, false>(S, L, nullptr);
}
template <int LEVELS, bool COUNT, int BLOCK>
__global__ __launch_bounds__(BLOCK) RENDER_OCCUPANCY void render_batch_kernel(RtDeviceScene S, RtLaunch L, RtBatch B) {
    render_body<LEVELS, COUNT, BLOCK, true>(S, L, &B);
}
// Blends the staged samples of a batch into the target, frame after frame per pixel (App.cpp:49,102-105).
__global__ __launch_bounds__(256) void blend_batch_kernel(RtLaunch L, RtBatch B) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= L.width * L.n_local_rows) return;
    const uint32_t x = i % L.width, lr = i / L.width;
    const uint32_t y = ((lr / L.band_rows) * L.band_parts + L.band_part) * L.band_rows + (lr % L.band_rows);
    if (y >= L.height) return;
    const uint32_t part_lr = ((lr / L.band_rows) * L.sub_parts + L.sub_part) * L.band_rows + (lr % L.band_rows);
    float4 *dstp = reinterpret_cast<float4 *>(L.color) + (L.compact_rows ? (size_t)part_lr * L.width + x : (size_t)y * L.width + x);
    float4 dst = *dstp;
    for (uint32_t f = 0; f < B.n; f++) {
        const float4 c = B.stage[(size_t)f * B.stage_stride + i];
        dst = blend_sample(dst, F3(c.x, c.y, c.z), B.accum_index[f], B.blend[f]);
    }
    *dstp = dst;
}

// Sorts the workgroup units by measured cost, most expensive first, into `order` (a counting sort over 128 logarithmic
// cost classes: four per octave) and clears the costs for the next launch.  One workgroup of 1024 threads.
__global__ __launch_bounds__(1024) void order_kernel(uint32_t *__restrict__ cost, uint32_t *__restrict__ order, uint32_t n) {
    __shared__ uint32_t count[128], start[128];
    if (threadIdx.x < 128) count[threadIdx.x] = 0;
    __syncthreads();
    auto cls = [](uint32_t c) -> uint32_t {
        if (c < 4u) return c;
        const uint32_t e = 31u - (uint32_t)__clz((int)c);
        return min(127u, e * 4u + ((c >> (e - 2u)) & 3u));
    };
    for (uint32_t i = threadIdx.x; i < n; i += 1024u) atomicAdd(&count[cls(cost[i])], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (int b = 127; b >= 0; b--) { start[b] = acc; acc += count[b]; }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 1024u) {
        order[atomicAdd(&start[cls(cost[i])], 1u)] = i;
        cost[i] = 0u;
    }
}

// ================================================================================================
// Pipeline 2: per-depth wavefront.  Path segments of one depth live in slot-indexed planes.  The slot space is cut
// into ranges of RANGE slots, each owned by one workgroup for the whole frame: at depth 0 a range holds four 16x16
// pixel tiles; the paths of a range that survive a depth are compacted to the front of the same range of the
// next-depth planes (a workgroup-local prefix sum: no global atomics), so range b holds live[depth][b] segments.
//   spheres_kernel   once per frame, all first segments, dense: ray generation, the 84-sphere loop and the root box test;
//                    the slots of segments that enter the octree are appended to the global traversal queue with
//                    one atomic per workgroup chunk
// and then for every depth:
//   traverse_kernel  persistent waves take batches of the traversal queue; a lane that empties its stack hands its
//                    result back and the wave refills its idle lanes together once enough of them wait
//   shade_kernel     all segments, dense: cylinder, portals, hit assembly, scattering; finished paths are tonemapped
//                    and blended into the target, continuing ones are compacted into the next depth's planes -- with
//                    their sphere loop and root box test done on the spot, and queued for the next traversal
// Counters (uint32): ctr[QT_COUNT + (d&1)], ctr[QT_HEAD + (d&1)] traversal queue of depth d (zeroed for depth d + 1 by
// the traversal kernel of depth d).
// ================================================================================================
enum { QT_COUNT = 0, QT_HEAD = 2 };
#define RANGE 1024u            // slots per workgroup-owned range (4 chunks of 256)
#ifndef REFILL_IDLE
#define REFILL_IDLE 16         // lanes that must be waiting before a traversal wave refills them together
#endif
#ifndef TRAVERSE_BATCH
#define TRAVERSE_BATCH 128u    // queue entries a traversal wave reserves with one atomic
#endif
#ifndef TRAVERSE_WG_PER_CU
#define TRAVERSE_WG_PER_CU 5   // persistent traversal workgroups per CU
#endif

// Workgroup-wide exclusive prefix of a per-lane predicate over the 4 waves of a 256-thread workgroup.
// Returns the lane's rank among the set lanes; total = number of set lanes.  Must be reached by all 256 threads.
DEVI uint32_t block_rank(bool pred, uint32_t *lds4, uint32_t &total) {
    const unsigned long long m = __ballot(pred);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    __syncthreads();                       // lds4 may still be read from the previous use
    if (lane == 0) lds4[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    const uint32_t c0 = lds4[0], c1 = lds4[1], c2 = lds4[2], c3 = lds4[3];
    total = c0 + c1 + c2 + c3;
    const uint32_t before = wave == 0 ? 0u : (wave == 1 ? c0 : (wave == 2 ? c0 + c1 : c0 + c1 + c2));
    return before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}

// The same for two predicates at once (one pair of barriers): lds8 holds the 4 + 4 per-wave counts.
DEVI void block_rank2(bool pa, bool pb, uint32_t *lds8, uint32_t &rank_a, uint32_t &total_a, uint32_t &rank_b, uint32_t &total_b) {
    const unsigned long long ma = __ballot(pa), mb = __ballot(pb);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    __syncthreads();                       // lds8 may still be read from the previous use
    if (lane == 0) { lds8[wave] = (uint32_t)__popcll(ma); lds8[4u + wave] = (uint32_t)__popcll(mb); }
    __syncthreads();
    const uint32_t a0 = lds8[0], a1 = lds8[1], a2 = lds8[2], a3 = lds8[3], b0 = lds8[4], b1 = lds8[5], b2 = lds8[6], b3 = lds8[7];
    total_a = a0 + a1 + a2 + a3; total_b = b0 + b1 + b2 + b3;
    const unsigned long long below = (1ull << lane) - 1ull;
    rank_a = (wave == 0 ? 0u : (wave == 1 ? a0 : (wave == 2 ? a0 + a1 : a0 + a1 + a2))) + (uint32_t)__popcll(ma & below);
    rank_b = (wave == 0 ? 0u : (wave == 1 ? b0 : (wave == 2 ? b0 + b1 : b0 + b1 + b2))) + (uint32_t)__popcll(mb & below);
}
// The slots of a workgroup's range that enter the octree are collected in LDS and appended to the traversal queue with ONE
// atomic per workgroup: adds to one address serialise in the L2 at several nanoseconds each, and a 4K frame has 32 400 chunks.
DEVI void flush_queue(const uint32_t *lds_q, uint32_t n, uint32_t *lds_base, uint32_t *counter, uint32_t *__restrict__ qt) {
    if (n == 0) return;                    // workgroup-uniform
    if (threadIdx.x == 0) *lds_base = atomicAdd(counter, n);
    __syncthreads();                       // (also publishes the last chunk's entries of lds_q)
    const uint32_t base = *lds_base;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) qt[base + i] = lds_q[i];
}

// Per-frame constants of the spheres as seen from the camera: (oc.xyz, c) with oc = camera - centre and
// c = dot(oc,oc) - r*r, the values every primary ray would compute (frag:234,236).
__global__ void primary_sphere_kernel(RtLaunch L, float4 *__restrict__ prim) {
    const uint32_t i = threadIdx.x;
    if (i >= NUM_SPHERES) return;
    const float4 s = c_spheres[i];
    const f3 oc = F3(L.u.camera_position[0], L.u.camera_position[1], L.u.camera_position[2]) - F3(s.x, s.y, s.z);
    prim[i] = make_float4(oc.x, oc.y, oc.z, dot3(oc, oc) - s.w * s.w);
}

// The first segment of every path of the frame: ray generation, the sphere loop, the root box test.
// With a stack of G.n frames (same view, own `time` each: consecutive samples of a progressive accumulation) the launch covers
// G.n x the frame's tiles; the paths of frame f carry the pixel index of staging plane f, and nothing after this kernel knows
// that they belong to different frames.
__global__ __launch_bounds__(256) void spheres_kernel(RtDeviceScene S, RtLaunch L, RtWorkspace WS, uint32_t n_tiles, RtStack G) {
    __shared__ uint32_t lds4[4];
    __shared__ uint32_t lds_base;
    __shared__ uint32_t lds_q[RANGE];
    __shared__ SphereLds lds_sph;
    stage_spheres(&lds_sph, WS.prim);
    __syncthreads();
    uint32_t q_count = 0;      // workgroup-uniform: entries of lds_q
    float4 *__restrict__ A = WS.pos_seed[0];
    float4 *__restrict__ B = WS.dir_pix[0];
    const uint32_t lo = blockIdx.x * RANGE;
    const uint32_t count = min(RANGE, (n_tiles - blockIdx.x * (RANGE / 256u)) * 256u);
    const bool counting = L.counters != nullptr;
    uint32_t n_sph = 0;
    for (uint32_t off = 0; off < count; off += 256u) {
        const uint32_t slot = lo + off + threadIdx.x;
        const uint32_t tiles_x = (L.width + 15u) / 16u, frame_tiles = n_tiles / G.n;
        const uint32_t v = xcd_contiguous(blockIdx.x * (RANGE / 256u) + off / 256u, n_tiles), stacked = v / frame_tiles, tile = v % frame_tiles;
        PixelId p = pixel_of_tile(L, tile % tiles_x, tile / tiles_x, threadIdx.x);
        p.pi += (size_t)stacked * G.stride;
        const bool valid = p.active;
        Ray ray;
        float closest = __builtin_inff();
        uint32_t closest_i = 0;
        bool enter = false;
        float seed_base = 0.0f;
        ray.pos = F3(L.u.camera_position[0], L.u.camera_position[1], L.u.camera_position[2]);
        ray.dir = F3(0.0f, 0.0f, 1.0f);
        if (valid) {
            float nx, ny;
            pixel_ndc(L, p, nx, ny);
            rt_uniforms U = L.u;
            U.time = G.time[stacked];
            generate_ray(U, nx, ny, ray.dir, seed_base);
        }
        A[slot] = make_float4(ray.pos.x, ray.pos.y, ray.pos.z, seed_base);
        B[slot] = make_float4(ray.dir.x, ray.dir.y, ray.dir.z, u2f(valid ? (uint32_t)p.pi : 0xFFFFFFFFu));
        WS.color[0][slot] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
        if (valid) {
            ray.inv = rcp3(ray.dir);
            SPHERE_LOOP(true, lds_sph, ray, closest, closest_i, counting ? &n_sph : nullptr);
            enter = root_box_test(S, ray, closest);
        }
        WS.t[slot] = closest;
        WS.sphere[slot] = closest_i;
        WS.tri[slot] = 0u;
        uint32_t total;
        const uint32_t rank = block_rank(enter, lds4, total);
        if (enter) lds_q[q_count + rank] = slot;
        q_count += total;
    }
    flush_queue(lds_q, q_count, &lds_base, &WS.ctr[QT_COUNT], WS.qt);
    if (counting) {
        const uint32_t v = wave_sum(n_sph);
        if ((threadIdx.x & 63u) == 0 && v) atomicAdd(&L.counters[6], (unsigned long long)v);
    }
}

// Exact near-first order, frames with eye-coplanar triangles only (see hits_eye_coplanar): marks the queued first segments that
// must take the reference's order by setting the sign bit of their closest distance, which the traversal kernel reads as that.
__global__ __launch_bounds__(256) void mark_eye_kernel(RtDeviceScene S, RtLaunch L, RtWorkspace WS) {
    if (*L.eye_count == 0u) return;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= WS.ctr[QT_COUNT]) return;
    const uint32_t slot = WS.qt[i];
    const float4 a = WS.pos_seed[0][slot], b = WS.dir_pix[0][slot];
    Ray ray;
    ray.pos = F3(a.x, a.y, a.z);
    ray.dir = F3(b.x, b.y, b.z);
    ray.inv = rcp3(ray.dir);
    if (hits_eye_coplanar(S, L, ray)) WS.t[slot] = -WS.t[slot];
}

template <int LEVELS, bool COUNT, int EXACT>
__global__ __launch_bounds__(256) void traverse_kernel(RtDeviceScene S, RtLaunch L, RtWorkspace WS, uint32_t depth) {
    __shared__ uint32_t lds_stack[LEVELS * 256];
    __shared__ TopLevels lds_top;
    stage_top_levels(&lds_top, S, EXACT != 0);
    __syncthreads();
    Exact x;
    x.begin(0.0f, false);
    const bool want_slot = depth == 0u && L.hit_slot != nullptr;   // the primary hit's slot goes to the caller's plane
    LevelStack st;
    st.col = &lds_stack[threadIdx.x];
    st.sp = 0;
    const uint32_t cur = depth & 1u;
    const float4 *__restrict__ A = WS.pos_seed[cur];
    const float4 *__restrict__ B = WS.dir_pix[cur];
    uint32_t *ctr = WS.ctr;
    const uint32_t qt_count = ctr[QT_COUNT + cur];
    // the queue of the other parity: read last by the previous depth's traversal, filled next by the shade_kernel that follows this launch
    if (blockIdx.x == 0 && threadIdx.x == 0) { ctr[QT_COUNT + (cur ^ 1u)] = 0; ctr[QT_HEAD + (cur ^ 1u)] = 0; }
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    Counters cn = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    bool has = false, exhausted = false;       // exhausted, batch_next, batch_end are wave-uniform
    uint32_t batch_next = 0, batch_end = 0;
    uint32_t slot = 0, closest_tri = 0;
    float closest = 0.0f;
    bool tri_hit = false;
    Ray ray;
    ray.pos = ray.dir = ray.inv = F3(0.0f, 0.0f, 0.0f);
    Walk walk;
    walk.node.tri_start = walk.node.tri_count = walk.node.child_mask = walk.node.first_child = 0;
    walk.node.lo = walk.node.mid = walk.node.hi = F3(0.0f, 0.0f, 0.0f);
    walk.tri_i = walk.tri_end = 0; walk.key = 0;
    for (;;) {
        unsigned long long busy = __ballot(has);
        if (!exhausted && 64 - __popcll(busy) >= REFILL_IDLE) {
            if (batch_next == batch_end) {           // reserve the next batch of queue entries for this wave
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&ctr[QT_HEAD + cur], TRAVERSE_BATCH);
                base = __shfl(base, 0, 64);
                batch_next = base < qt_count ? base : qt_count;
                batch_end = base + TRAVERSE_BATCH < qt_count ? base + TRAVERSE_BATCH : qt_count;
                if (batch_next == batch_end) exhausted = true;
            }
            const uint32_t my = batch_next + (uint32_t)__popcll(~busy & lt_mask);
            if (!has && my < batch_end) {
                slot = WS.qt[my];
                const float4 a = A[slot], b = B[slot];
                ray.pos = F3(a.x, a.y, a.z);
                ray.dir = F3(b.x, b.y, b.z);
                ray.inv = rcp3(ray.dir);
                // (the sign bit: mark_eye_kernel's mark on a first segment that must take the reference's order, see hits_eye_coplanar;
                // the distance a segment is queued with is the sphere loop's, strictly positive or infinite: sphere_accept)
                const float ct = WS.t[slot];
                const bool marked = EXACT != 0 && (f2u(ct) >> 31) != 0u;
                closest = __builtin_fabsf(ct);
                closest_tri = 0; tri_hit = false;
                walk.begin(st, order_key(ray, L.near_first != 0u));
                if (EXACT) {
                    x.begin(closest, want_slot);
                    if (hits_risky(S, &lds_top, ray) || marked) { x.nf = false; walk.begin(st, 0u); if (COUNT) cn.fallbacks++; }
                }
                has = true;
                if (COUNT) cn.entered++;
            }
            const uint32_t taken = min(64u - (uint32_t)__popcll(busy), batch_end - batch_next);
            batch_next += taken;
            busy = __ballot(has);
        }
        if (busy == 0) { if (exhausted) break; else continue; }
        const bool was = has;
        walk_step<COUNT, EXACT>(S, &lds_top, ray, st, walk, x, L.near_first != 0u, closest, closest_tri, tri_hit, has, cn);
        if (was && !has) {
            WS.t[slot] = closest;
            if (tri_hit) WS.tri[slot] = closest_tri | 0x80000000u;
        }
    }
    if (COUNT && L.counters) flush_counters(L.counters, cn);
}

template <bool D0, bool COUNT>
__global__ __launch_bounds__(256) void shade_kernel(RtDeviceScene S, RtLaunch L, RtWorkspace WS, uint32_t depth, uint32_t n_tiles) {
    __shared__ uint32_t lds8[8];
    __shared__ uint32_t lds_base;
    __shared__ uint32_t lds_q[RANGE];
    __shared__ PortalBasis lds_portals[2];
    __shared__ SphereLds lds_sph;
    stage_spheres(&lds_sph, c_spheres);      // (published by the barrier inside stage_portals)
    stage_portals(lds_portals, L.u);
    const uint32_t cur = depth & 1u, nxt = cur ^ 1u;
    const float4 *__restrict__ A = WS.pos_seed[cur];
    const float4 *__restrict__ B = WS.dir_pix[cur];
    const float4 *__restrict__ C = WS.color[cur];
    uint32_t *ctr = WS.ctr;
    const bool counting = L.counters != nullptr;
    uint32_t n_sph = 0;
    const uint32_t lo = blockIdx.x * RANGE;
    const uint32_t count = D0 ? min(RANGE, (n_tiles - blockIdx.x * (RANGE / 256u)) * 256u) : WS.live[cur][blockIdx.x];
    uint32_t out_count = 0;    // workgroup-uniform: survivors of this range so far
    uint32_t q_count = 0;      // workgroup-uniform: of them, those that enter the octree (entries of lds_q)
    Counters cn = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t off = 0; off < count; off += 256u) {
        const uint32_t slot = lo + off + threadIdx.x;
        bool valid = off + threadIdx.x < count;
        bool cont = false;
        float4 na = make_float4(0, 0, 0, 0), nb = na, nc = na;
        if (valid) {
            const float4 a = A[slot], b = B[slot];
            const uint32_t pix = f2u(b.w);
            valid = pix != 0xFFFFFFFFu;
            if (valid) {
                const float4 c = C[slot];
                Ray ray;
                ray.pos = F3(a.x, a.y, a.z);
                ray.dir = F3(b.x, b.y, b.z);
                ray.inv = rcp3(ray.dir);
                Seed seed;
                seed.base = a.w; seed.k = 0;
                f3 color = F3(c.x, c.y, c.z);
                const float closest = WS.t[slot];
                const uint32_t closest_i = WS.sphere[slot];
                const uint32_t tri_word = WS.tri[slot];
                if (COUNT) cn.rays++;
                const Hit h = finish_intersection<COUNT>(S, L.u, lds_portals, ray, closest, closest_i, tri_word & 0x7FFFFFFFu, (tri_word >> 31) != 0, cn);
                if (D0) {
                    if (L.depth) L.depth[pix] = first_hit_depth(L.u, h, ray);
                    if (L.hit_kind) L.hit_kind[pix] = h.kind;
                    if (L.hit_slot) L.hit_slot[pix] = h.slot;
                    if (L.hit_t) L.hit_t[pix] = h.t;
                }
                bool done = false;
                if (!h.has_hit) {
                    color = color * sample_sky(S, ray.dir);
                    done = true;
                } else {
                    done = depth + 1u >= L.u.max_recursion_limit;
                    scatter(L.u, h, ray, color, seed, done);
                }
                if (done) {
                    write_pixel(L, pix, tonemap(color));
                    if (L.n_rays) L.n_rays[pix] = depth + 1u;
                } else {
                    cont = true;
                    na = make_float4(ray.pos.x, ray.pos.y, ray.pos.z, seed.base);
                    nb = make_float4(ray.dir.x, ray.dir.y, ray.dir.z, b.w);
                    nc = make_float4(color.x, color.y, color.z, 0.0f);
                }
            }
        }
        // The next segment's sphere loop and root box test run here, on the ray still in registers (what spheres_kernel does for
        // the first segment): its planes are written once, and the segments that enter the octree go to the next depth's queue.
        float nclosest = __builtin_inff();
        uint32_t nclosest_i = 0;
        bool enter = false;
        if (cont) {
            Ray nr;
            nr.pos = F3(na.x, na.y, na.z);
            nr.dir = F3(nb.x, nb.y, nb.z);
            nr.inv = rcp3(nr.dir);
            SPHERE_LOOP(false, lds_sph, nr, nclosest, nclosest_i, counting ? &n_sph : nullptr);
            enter = root_box_test(S, nr, nclosest);
        }
        uint32_t total, rank, total_e, rank_e;
        block_rank2(cont, enter, lds8, rank, total, rank_e, total_e);
        const uint32_t ni = lo + out_count + rank;       // never ahead of the slots this workgroup still has to read (all of this
        if (cont) {                                      // chunk's were read before the barriers of block_rank2)
            WS.pos_seed[nxt][ni] = na;
            WS.dir_pix[nxt][ni] = nb;
            WS.color[nxt][ni] = nc;
            WS.t[ni] = nclosest;
            WS.sphere[ni] = nclosest_i;
            WS.tri[ni] = 0u;
        }
        out_count += total;
        if (enter) lds_q[q_count + rank_e] = ni;
        q_count += total_e;
    }
    flush_queue(lds_q, q_count, &lds_base, &ctr[QT_COUNT + nxt], WS.qt);
    if (threadIdx.x == 0) WS.live[nxt][blockIdx.x] = out_count;
    if (counting) {
        const uint32_t v = wave_sum(n_sph);
        if ((threadIdx.x & 63u) == 0 && v) atomicAdd(&L.counters[6], (unsigned long long)v);
    }
    if (COUNT && L.counters) flush_counters(L.counters, cn);
}


// ================================================================================================
// Pipeline 3: persistent path kernel.  A fixed number of waves stays resident for the whole frame; every lane carries
// one path at a time and is in one of two phases: traversing the octree (a stack is live), or waiting for the
// "segment block" -- hit assembly, scattering or sky, pixel write when the path ends, a new pixel from the wave's
// supply when the lane is free, then the sphere loop and the root box test of the next segment.  A wave runs the
// segment block for all waiting lanes together once PATH_REFILL of them wait (or none traverses), and otherwise
// advances the traversing lanes node by node, so neither phase executes with a nearly empty wave and the serial
// chain of one long ray overlaps everything else the machine does.  Pixels are handed out in 8x8 tiles: a wave's
// tiles come from global counters (one atomic per 64 pixels, spread over PATH_POOLS counters).
// ================================================================================================
#ifndef PATH_REFILL
#define PATH_REFILL 16         // waiting lanes that trigger the segment block while others still traverse
#endif
#ifndef PATH_MIN_VISITS
#define PATH_MIN_VISITS 4      // node visits the traversing lanes get between two segment blocks
#endif
#define PATH_POOLS 64u         // tile counters (atomics on one address serialise at several ns each: spread them)
#define PATH_POOL_STRIDE 64u   // words between two counters: one per 256-byte line
#ifndef PATH_TAIL_VISITS
#define PATH_TAIL_VISITS 16    // the same once the wave's pixel supply is dry (any waiting lane then triggers a block)
#endif
#ifndef PATH_WG_PER_CU
#define PATH_WG_PER_CU 5       // resident workgroups per CU (= PATH_WAVES_PER_SIMD: 4 waves per workgroup, 4 SIMDs per CU)
#endif

#ifndef PATH_WAVES_PER_SIMD
#define PATH_WAVES_PER_SIMD 5
#endif
template <int LEVELS, bool COUNT, bool EXACT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LEVELS <= 20 ? PATH_WAVES_PER_SIMD : 3, LEVELS <= 20 ? PATH_WAVES_PER_SIMD : 3))) void path_kernel(RtDeviceScene S, RtLaunch L, uint32_t *pool, uint32_t n_tiles) {
    __shared__ uint32_t lds_stack[LEVELS * 256];
    __shared__ SphereLds lds_sph;
    __shared__ PortalBasis lds_portals[2];
    __shared__ TopLevels lds_top;
    stage_top_levels(&lds_top, S, EXACT != 0);
    stage_spheres(&lds_sph, c_spheres);
    stage_portals(lds_portals, L.u);
    Exact x;
    x.begin(0.0f, false);
    LevelStack st;
    st.col = &lds_stack[threadIdx.x];
    st.sp = 0;
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint32_t n_units = n_tiles * 4u;   // unit = 8x8 pixels = quarter of a 16x16 tile
    const uint32_t limit = L.u.max_recursion_limit;
    // Wave-uniform pixel supply.  Counter c hands out the units c, c + PATH_POOLS, c + 2 PATH_POOLS, ...; a wave draws
    // from its home counter and moves on to the next one when that runs dry, until all of them have.
    uint32_t home = (blockIdx.x * 4u + (threadIdx.x >> 6)) % PATH_POOLS, dry = 0;
    auto next_unit = [&]() -> uint32_t {
        while (dry < PATH_POOLS) {
            uint32_t k = 0;
            if (lane == 0) k = atomicAdd(&pool[home * PATH_POOL_STRIDE], 1u);
            const uint32_t u = __shfl(k, 0, 64) * PATH_POOLS + home;
            if (u < n_units) return u;
            home = (home + 1u) % PATH_POOLS;
            dry++;
        }
        return 0xFFFFFFFFu;
    };
#ifdef PATH_TRACE
    const unsigned long long t_start = wall_clock64();
#endif
    uint32_t unit = limit > 0 ? next_unit() : 0xFFFFFFFFu, cursor = 0;   // `cursor` pixels of `unit` are handed out
    Counters cn = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    bool has_path = false, traversing = false, tri_hit = false;
    Walk walk;
    walk.node.tri_start = walk.node.tri_count = walk.node.child_mask = walk.node.first_child = 0;
    walk.node.lo = walk.node.mid = walk.node.hi = F3(0.0f, 0.0f, 0.0f);
    walk.tri_i = walk.tri_end = 0; walk.key = 0;
    uint32_t pix = 0, seg = 0, closest_i = 0, closest_tri = 0;
    float closest = 0.0f;
    Ray ray;
    ray.pos = ray.dir = ray.inv = F3(0.0f, 0.0f, 0.0f);
    f3 color = F3(1.0f, 1.0f, 1.0f);
    Seed seed;
    seed.base = 0.0f; seed.k = 0;
    for (;;) {
        const unsigned long long trav = __ballot(traversing);
        const unsigned long long work = __ballot(!traversing && (has_path || (unit < n_units && limit > 0)));
        if (trav == 0 && work == 0) break;
        // once the wave's pixel supply is dry only latency matters: any waiting lane gets the block at once
        const uint32_t refill = unit < n_units ? PATH_REFILL : 1u;
        if (work != 0 && (trav == 0 || (uint32_t)__popcll(work) >= refill)) {
            if (has_path && !traversing) {   // the segment's intersection is complete (frag:535-563,575-664)
                const Hit h = finish_intersection<COUNT>(S, L.u, lds_portals, ray, closest, closest_i, closest_tri, tri_hit, cn);
                if (seg == 0) {
                    if (L.depth) L.depth[pix] = first_hit_depth(L.u, h, ray);
                    if (L.hit_kind) L.hit_kind[pix] = h.kind;
                    if (L.hit_slot) L.hit_slot[pix] = h.slot;
                    if (L.hit_t) L.hit_t[pix] = h.t;
                }
                seg++;
                bool done;
                if (!h.has_hit) {
                    color = color * sample_sky(S, ray.dir);
                    done = true;
                } else {
                    done = seg >= limit;
                    scatter(L.u, h, ray, color, seed, done);
                    if (!done) ray.inv = rcp3(ray.dir);
                }
                if (done) {
                    write_pixel(L, pix, tonemap(color));
                    if (L.n_rays) L.n_rays[pix] = seg;
                    has_path = false;
                }
            }
            const unsigned long long need = __ballot(!has_path);
            bool fresh = false;
            if (need != 0 && unit < n_units && limit > 0) {   // wave-uniform
                uint32_t tile_x, tile_y;
                path_tile(L, unit >> 2, tile_x, tile_y);
                const uint32_t n_take = min((uint32_t)__popcll(need), 64u - cursor);
                const uint32_t rank = (uint32_t)__popcll(need & lt_mask);
                if (!has_path && rank < n_take) {
                    const PixelId p = pixel_of_tile(L, tile_x, tile_y, (unit & 3u) * 64u + cursor + rank);
                    if (p.active) {
                        float nx, ny;
                        pixel_ndc(L, p, nx, ny);
                        ray.pos = F3(L.u.camera_position[0], L.u.camera_position[1], L.u.camera_position[2]);
                        generate_ray(L.u, nx, ny, ray.dir, seed.base);
                        seed.k = 0;
                        ray.inv = rcp3(ray.dir);
                        color = F3(1.0f, 1.0f, 1.0f);
                        pix = (uint32_t)p.pi;
                        seg = 0;
                        has_path = true;
                        fresh = true;
                    }
                }
                cursor += n_take;
                if (cursor == 64u) {
                    cursor = 0;
                    unit = next_unit();
                }
            }
            (void)fresh;
            if (has_path && !traversing) {   // next segment: spheres, then the octree if its root box is hit (frag:437-453)
                closest = __builtin_inff();
                closest_i = 0; closest_tri = 0; tri_hit = false;
                if (COUNT) cn.rays++;
                SPHERE_LOOP(false, lds_sph, ray, closest, closest_i, COUNT ? &cn.sph_tests : nullptr);
                if (root_box_test(S, ray, closest)) {
                    if (COUNT) cn.entered++;
                    walk.begin(st, order_key(ray, L.near_first != 0u));
                    if (EXACT) {
                        x.begin(closest, seg == 0u && L.hit_slot != nullptr);
                        if (hits_risky(S, &lds_top, ray) || (seg == 0u && hits_eye_coplanar(S, L, ray))) { x.nf = false; walk.begin(st, 0u); if (COUNT) cn.fallbacks++; }
                    }
                    traversing = true;
                }
            }
        }
        for (uint32_t steps = 0;; steps++) {
            const unsigned long long t = __ballot(traversing);
            if (t == 0) break;
            if (steps >= (unit < n_units ? PATH_MIN_VISITS : PATH_TAIL_VISITS)) {
                const unsigned long long w = __ballot(!traversing && (has_path || unit < n_units));
                if ((uint32_t)__popcll(w) >= (unit < n_units ? PATH_REFILL : 1u)) break;
            }
            walk_step<COUNT, EXACT ? 2 : 0>(S, &lds_top, ray, st, walk, x, L.near_first != 0u, closest, closest_tri, tri_hit, traversing, cn);
        }
    }
#ifdef PATH_TRACE   // development aid: per-wave start / end wall clock behind the counters
    if (lane == 0) {
        const uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6);
        unsigned long long *tr = reinterpret_cast<unsigned long long *>(pool + PATH_POOLS * PATH_POOL_STRIDE);
        tr[2 * w] = t_start; tr[2 * w + 1] = wall_clock64();
    }
#endif
    if (COUNT && L.counters) flush_counters(L.counters, cn);
}

template <int LEVELS>
void launch_path(const RtDeviceScene &S, const RtLaunch &L, uint32_t *pool, uint32_t n_tiles, hipStream_t st, dim3 grid) {
    if (L.near_first == 2u) {
        if (L.counters) hipLaunchKernelGGL((path_kernel<LEVELS, true, true>), grid, dim3(256), 0, st, S, L, pool, n_tiles);
        else hipLaunchKernelGGL((path_kernel<LEVELS, false, true>), grid, dim3(256), 0, st, S, L, pool, n_tiles);
    } else {
        if (L.counters) hipLaunchKernelGGL((path_kernel<LEVELS, true, false>), grid, dim3(256), 0, st, S, L, pool, n_tiles);
        else hipLaunchKernelGGL((path_kernel<LEVELS, false, false>), grid, dim3(256), 0, st, S, L, pool, n_tiles);
    }
}

template <int LEVELS>
int launch_megakernel(const RtDeviceScene &S, const RtLaunch &L, hipStream_t st, dim3 grid, const RtBatch *batch) {
    if (batch) {
        // one workgroup per 8x8 tile and group of RT_BATCH_BLOCK / 64 frames
        grid.x *= (RENDER_BLOCK / 64u) * rt_batch_groups(batch->n);
        if (L.counters) hipLaunchKernelGGL((render_batch_kernel<LEVELS, true, RT_BATCH_BLOCK>), grid, dim3(RT_BATCH_BLOCK), 0, st, S, L, *batch);
        else hipLaunchKernelGGL((render_batch_kernel<LEVELS, false, RT_BATCH_BLOCK>), grid, dim3(RT_BATCH_BLOCK), 0, st, S, L, *batch);
        return (int)hipGetLastError();
    }
    if (L.counters) hipLaunchKernelGGL((render_kernel<LEVELS, true, RENDER_BLOCK>), grid, dim3(RENDER_BLOCK), 0, st, S, L);
    else hipLaunchKernelGGL((render_kernel<LEVELS, false, RENDER_BLOCK>), grid, dim3(RENDER_BLOCK), 0, st, S, L);
    return (int)hipGetLastError();
}

template <int LEVELS>
void launch_traverse(const RtDeviceScene &S, const RtLaunch &L, const RtWorkspace &WS, uint32_t d, hipStream_t st, dim3 grid) {
    if (L.near_first == 2u && d == 0u && L.hit_slot != nullptr) {      // the primary hit's slot goes to the caller's plane
        if (L.counters) hipLaunchKernelGGL((traverse_kernel<LEVELS, true, 2>), grid, dim3(256), 0, st, S, L, WS, d);
        else hipLaunchKernelGGL((traverse_kernel<LEVELS, false, 2>), grid, dim3(256), 0, st, S, L, WS, d);
    } else if (L.near_first == 2u) {
        if (L.counters) hipLaunchKernelGGL((traverse_kernel<LEVELS, true, 1>), grid, dim3(256), 0, st, S, L, WS, d);
        else hipLaunchKernelGGL((traverse_kernel<LEVELS, false, 1>), grid, dim3(256), 0, st, S, L, WS, d);
    } else {
        if (L.counters) hipLaunchKernelGGL((traverse_kernel<LEVELS, true, 0>), grid, dim3(256), 0, st, S, L, WS, d);
        else hipLaunchKernelGGL((traverse_kernel<LEVELS, false, 0>), grid, dim3(256), 0, st, S, L, WS, d);
    }
}

}  // namespace

// The persistent kernel's block: the tile counters, (development builds: wave time stamps,) then the frame's eye-coplanar
// triangles -- a count in the first of four words, RT_EYE_CAP entries.  Zeroed by every launch.
#ifdef PATH_TRACE
#define PATH_EYE_WORDS ((size_t)PATH_POOLS * PATH_POOL_STRIDE + 16384 * 4)
#else
#define PATH_EYE_WORDS ((size_t)PATH_POOLS * PATH_POOL_STRIDE)
#endif
size_t rt_pool_bytes() { return (PATH_EYE_WORDS + 4 + RT_EYE_CAP * 4) * sizeof(uint32_t); }
// Exhaustive comparison of rcp_ieee / sqrt_ieee with the compiler's correctly rounded sequences: op 0: 1/x, op 1: sqrt(x),
// over all 2^32 bit patterns.  out[0] = number of inputs whose results differ in any bit (NaN results compare equal to
// NaN results), out[1] = the first such input found.
namespace {
__global__ __launch_bounds__(256) void math_check_kernel(int op, unsigned long long *out) {
    const uint32_t base = (blockIdx.x * 256u + threadIdx.x) * 256u;
    uint32_t bad = 0, first = 0;
    for (uint32_t k = 0; k < 256u; k++) {
        const float x = u2f(base + k);
        const float a = op == 0 ? rcp_ieee(x) : sqrt_ieee(x), b = op == 0 ? rcp_slow(x) : sqrt_slow(x);
        const bool same = (a != a && b != b) || f2u(a) == f2u(b);
        if (!same) { if (!bad) first = base + k; bad++; }
    }
    if (bad) { atomicAdd(&out[0], (unsigned long long)bad); atomicMax(&out[1], (unsigned long long)first); }
}
}  // namespace
int rt_math_check(int op, unsigned long long out[2]) {
    unsigned long long *d = nullptr;
    hipError_t e = hipMalloc((void **)&d, 16);
    if (e == hipSuccess) e = hipMemset(d, 0, 16);
    if (e == hipSuccess) { hipLaunchKernelGGL(math_check_kernel, dim3(1u << 16), dim3(256), 0, nullptr, op, d); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpy(out, d, 16, hipMemcpyDeviceToHost);
    if (d) (void)hipFree(d);
    return (int)e;
}

size_t rt_render_units(uint32_t width, uint32_t n_local_rows) {
    return (size_t)((width + 15u) / 16u) * ((n_local_rows + 15u) / 16u);
}
size_t rt_workspace_ranges(uint32_t width, uint32_t n_local_rows, uint32_t stack) {
    const size_t n_tiles = (size_t)((width + 15u) / 16u) * ((n_local_rows + 15u) / 16u) * stack;
    return (n_tiles + RANGE / 256u - 1) / (RANGE / 256u);
}
size_t rt_workspace_slots(uint32_t width, uint32_t n_local_rows, uint32_t stack) { return rt_workspace_ranges(width, n_local_rows, stack) * RANGE; }

namespace {
// Blends the staged sample planes of a stack of frames (each laid out like a compact target of the rank's share, G.stride
// pixels apart) into the colour target in frame order, with every frame's own accumulation index and blend state, for the
// pixels of the launch's sub-frame (App.cpp:49,102-105).
__global__ __launch_bounds__(256) void blend_stage_kernel(RtLaunch L, const float4 *__restrict__ stage, RtStack G) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= L.width * L.n_local_rows) return;
    const uint32_t x = i % L.width, lr = i / L.width;
    const uint32_t y = ((lr / L.band_rows) * L.band_parts + L.band_part) * L.band_rows + (lr % L.band_rows);
    if (y >= L.height) return;
    const uint32_t part_lr = ((lr / L.band_rows) * L.sub_parts + L.sub_part) * L.band_rows + (lr % L.band_rows);
    const size_t si = (size_t)part_lr * L.width + x;
    float4 *dstp = reinterpret_cast<float4 *>(L.color) + (L.compact_rows ? si : (size_t)y * L.width + x);
    float4 dst = *dstp;   // (read even when the first frame does not blend: one code path, and the target is mapped memory anyway)
    for (uint32_t f = 0; f < G.n; f++) {
        const float4 c = stage[(size_t)f * G.stride + si];
        dst = blend_sample(dst, F3(c.x, c.y, c.z), G.accum_index[f], G.blend[f]);
    }
    *dstp = dst;
}
}
int rt_launch_blend(const RtLaunch &L, const float4 *stage, const RtStack &stack, void *stream) {
    const uint32_t px = L.width * L.n_local_rows;
    if (px == 0 || stack.n == 0) return 0;
    hipLaunchKernelGGL(blend_stage_kernel, dim3((px + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, L, stage, stack);
    return (int)hipGetLastError();
}

int rt_launch_render(const RtDeviceScene &S, const RtLaunch &L_in, const RtWorkspace *WS, uint32_t *pool, void *stream, const char **err, RtProbe *probe, const RtBatch *batch,
                     const RtStack *stack) {
    RtLaunch L = L_in;
    L.eye_count = nullptr; L.eye_list = nullptr;
    const uint32_t tiles_x = (L.width + 15u) / 16u, tiles_y = (L.n_local_rows + 15u) / 16u;
    if (tiles_x == 0 || tiles_y == 0) return 0;
    const uint32_t n_tiles = tiles_x * tiles_y, frame_tiles = n_tiles;
    if (stack && (!WS || stack->n < 1 || stack->n > RT_MAX_STACK)) { if (err) *err = "stacked frames are a feature of the wavefront pipeline"; return 1; }
    hipStream_t st = (hipStream_t)stream;
    // the traversal stack holds at most one entry per tree level (the root's virtual level included)
    const uint32_t need = S.max_depth + 2u;
    int rc = 0;
    const bool timed = probe && probe->used < probe->capacity;
    auto mark = [&](uint32_t which) { if (timed) (void)hipEventRecord(probe->events[2 * probe->used + which], st); };
    if (pool) {
        const uint32_t n_cu = L.n_cu ? L.n_cu : 256u;
        const uint32_t n_units = n_tiles * 4u;
        const uint32_t wgs = min(n_cu * PATH_WG_PER_CU, (n_units + 3u) / 4u);
        (void)hipMemsetAsync(pool, 0, rt_pool_bytes(), st);
        if (L.near_first == 2u && S.n_planes) {
            L.eye_count = pool + PATH_EYE_WORDS; L.eye_list = reinterpret_cast<const float4 *>(pool + PATH_EYE_WORDS + 4);
            hipLaunchKernelGGL(eye_planes_kernel, dim3((S.n_planes + 255u) / 256u), dim3(256), 0, st, S.planes, S.n_planes, L.u.camera_position[0], L.u.camera_position[1], L.u.camera_position[2],
                               pool + PATH_EYE_WORDS, reinterpret_cast<float4 *>(pool + PATH_EYE_WORDS + 4));
        }
        mark(0);
        if (need <= 12) launch_path<12>(S, L, pool, n_tiles, st, dim3(wgs));
        else if (need <= 20) launch_path<20>(S, L, pool, n_tiles, st, dim3(wgs));
        else launch_path<RT_MAX_TREE_DEPTH + 2>(S, L, pool, n_tiles, st, dim3(wgs));
        mark(1);
        if (timed) probe->launches[probe->used++] = 1;
        rc = (int)hipGetLastError();
    } else if (!WS) {
        const dim3 grid(n_tiles);
        mark(0);
        if (need <= 12) rc = launch_megakernel<12>(S, L, st, grid, batch);
        else if (need <= 19) rc = launch_megakernel<19>(S, L, st, grid, batch);   // (the reference's depth limit of 18 gives 19)
        else if (need <= 20) rc = launch_megakernel<20>(S, L, st, grid, batch);
        else rc = launch_megakernel<RT_MAX_TREE_DEPTH + 2>(S, L, st, grid, batch);
        mark(1);
        if (batch && rt_batch_groups(batch->n) > 1u) hipLaunchKernelGGL(blend_batch_kernel, dim3((L.width * L.n_local_rows + 255u) / 256u), dim3(256), 0, st, L, *batch);
        if (L.cost && L.order_next) hipLaunchKernelGGL(order_kernel, dim3(1), dim3(1024), 0, st, L.cost, L.order_next, n_tiles);
        if (timed) probe->launches[probe->used++] = 1;
    } else {
        const bool count = L.counters != nullptr;
        RtStack G{};
        if (stack) G = *stack;
        else { G.n = 1; G.stride = 0; G.time[0] = L.u.time; }
        const uint32_t n_tiles = frame_tiles * G.n;
        const dim3 grid((uint32_t)rt_workspace_ranges(L.width, L.n_local_rows, G.n));   // one workgroup per slot range
        const dim3 tgrid((L.n_cu ? L.n_cu : 256u) * TRAVERSE_WG_PER_CU);             // persistent traversal workgroups
        (void)hipMemsetAsync(WS->ctr, 0, 8 * sizeof(uint32_t), st);
        hipLaunchKernelGGL(primary_sphere_kernel, dim3(1), dim3(128), 0, st, L, WS->prim);
        if (L.near_first == 2u && S.n_planes) {
            L.eye_count = WS->ctr + 4; L.eye_list = WS->prim + 96;
            hipLaunchKernelGGL(eye_planes_kernel, dim3((S.n_planes + 255u) / 256u), dim3(256), 0, st, S.planes, S.n_planes, L.u.camera_position[0], L.u.camera_position[1], L.u.camera_position[2],
                               WS->ctr + 4, WS->prim + 96);
        }
        // (every depth is enqueued: a depth no path reaches costs three launches that find empty ranges and an empty queue,
        // whereas asking the device whether any path is left would block the caller inside an enqueue)
        const uint32_t limit = L.u.max_recursion_limit;
        for (uint32_t d = 0; d < limit; d++) {
            if (d == 0) hipLaunchKernelGGL(spheres_kernel, grid, dim3(256), 0, st, S, L, *WS, n_tiles, G);    // (later depths: inside the previous shade_kernel)
            if (d == 0 && L.eye_count && S.n_nodes) hipLaunchKernelGGL(mark_eye_kernel, dim3(grid.x * (RANGE / 256u)), dim3(256), 0, st, S, L, *WS);
            if (S.n_nodes) {
                const bool t = timed && probe->used < probe->capacity;
                if (t) mark(0);
                if (need <= 12) launch_traverse<12>(S, L, *WS, d, st, tgrid);
                else if (need <= 20) launch_traverse<20>(S, L, *WS, d, st, tgrid);
                else launch_traverse<RT_MAX_TREE_DEPTH + 2>(S, L, *WS, d, st, tgrid);
                if (t) { mark(1); probe->launches[probe->used++] = 1; }
            }
            if (d == 0) {
                if (count) hipLaunchKernelGGL((shade_kernel<true, true>), grid, dim3(256), 0, st, S, L, *WS, d, n_tiles);
                else hipLaunchKernelGGL((shade_kernel<true, false>), grid, dim3(256), 0, st, S, L, *WS, d, n_tiles);
            } else {
                if (count) hipLaunchKernelGGL((shade_kernel<false, true>), grid, dim3(256), 0, st, S, L, *WS, d, n_tiles);
                else hipLaunchKernelGGL((shade_kernel<false, false>), grid, dim3(256), 0, st, S, L, *WS, d, n_tiles);
            }
        }
        rc = (int)hipGetLastError();
    }
    if (rc != 0) { if (err) *err = hipGetErrorString((hipError_t)rc); return rc; }
    return 0;
}
