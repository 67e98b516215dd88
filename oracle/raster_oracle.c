/*
 * TEST INFRASTRUCTURE (oracle side) -- not part of the shipped product.
 *
 * CPU restatement of the raster pass that follows the ray-tracing draw in the reference's frame
 * (/root/reference/app/core/src/App.cpp:142-166): indexed triangle meshes through
 * app/assets/Vert_PosNormTex.vert:19-24 and app/assets/Frag_LightingSimple.frag:29-41, back faces culled
 * (App.cpp:50-51), depth test GL_LEQUAL with depth writes against the 24-bit attachment (App.cpp:54,
 * Framebuffer.cpp:10-12), blending off, one RGBA8 texture sampled with GL_REPEAT (GLUtils.cpp:211-224).
 *
 * The arithmetic is the one Mesa llvmpipe executes, found by probing it with single triangles (window
 * position, 1/w, varyings and texture lookups read back bit for bit):
 *   - vertex stage: viewProj * (world * v), each product ((c0*x + c1*y) + c2*z) + c3*w, not fused;
 *   - ndc = clip.xyz * (1 / clip.w); window = fma(ndc, scale, translate) with scale = translate = size / 2 (z: 0.5);
 *   - coverage: window x, y minus the half-pixel centre, snapped to 1/256 pixel (round to nearest); edge functions
 *     in integers at the pixel centres; a centre exactly on an edge belongs to the triangle on left and bottom edges;
 *   - planes: a counter-clockwise triangle (v0, v1, v2) is set up from (v1, v0, v2): with P = v1, Q = v0, R = v2,
 *     dx01 = P.x - Q.x, dx20 = R.x - P.x (float window coordinates, y alike), ooa = 1 / (dx01*dy20 - dx20*dy01),
 *     dadx = da01*(dy20*ooa) - da20*(dy01*ooa), dady = da20*(dx01*ooa) - da01*(dx20*ooa),
 *     a0 = a(P) - (dadx*(P.x - 0.5) + dady*(P.y - 0.5)), none of it fused;
 *   - per pixel (integer x, y): a = fma(dady, y, fma(dadx, x, a0)) for z and for 1/w; a varying is set up from
 *     value * (1/w) per vertex and evaluated as fma(...) * (1 / interpolated 1/w);
 *   - depth24 = round-to-nearest-even(float(clamp(z, 0, 1) * 16777215.0f));
 *   - fragment stage: dot(a, b) = (a.z*b.z + a.y*b.y) + a.x*b.x, normalize(v) = v * (1 / sqrt(dot(v, v))), clamp = min(max(., 0), 1),
 *     nothing fused;
 *   - texture: per axis, a power-of-two size gives (c * size) * 256, any other size takes the fraction first, f = min(c - floor(c),
 *     0.99999994), and gives (f * size) * 256; rounded to nearest even, minus 128: texel = value >> 8 (wrapped), weight = value & 255;
 *     8-bit lerp(a, b, w) = a + (((b - a) * w + 128) >> 8) along x, then along y (then between two mip levels); result * (1 / 255).
 *   - clipping: a triangle with a vertex outside any of the six planes -w <= x, y, z <= w is clipped plane by plane (x <= w, x >= -w,
 *     y <= w, y >= -w, z >= -w, z <= w; only the planes some vertex violates), walking the polygon from its first vertex; a new
 *     vertex on an edge is a + t * (b - a) for the clip position and every output, a being the end of the edge nearer to the
 *     plane (|d_a| < |d_b|, d = w -+ coordinate) whether it is inside or outside, t = d_a / (d_a - d_b), not fused, and its window position is ((clip * (1/w)) * scale) + translate, not fused either; the polygon (p0 .. pn-1) is
 *     drawn as the triangles (p[i-1], p[i], p0), each culled first by the sign of (v0 - v2) x (v1 - v2) in float window coordinates
 *     (the pieces of a clipped triangle only; negative = clockwise is dropped, zero passes on to the snapped-coordinate test).
 *   - mip levels and their use: see build_mip_chain and sample_texture below (level of detail per 2 x 2 quad from the differences
 *     of the texture coordinates at its lower left pixel, a log2 read off the float's bits, an 8-bit weight between two levels).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float pos[4]; float win[4]; float attr[8]; } out_vertex;   /* clip position; window x, y, z and 1/w; world position (3), normal (3), texcoord (2) */

static void mat_vec(const float *M, const float v[4], float out[4]) {
    for (int r = 0; r < 4; r++) out[r] = ((M[0 + r] * v[0] + M[4 + r] * v[1]) + M[8 + r] * v[2]) + M[12 + r] * v[3];
}

/* Vert_PosNormTex.vert:19-24 */
static void vertex_stage(const float *in /* pos3 normal3 tex2 */, const float *view_proj, const float *world, const float *world_it, out_vertex *o) {
    const float p[4] = {in[0], in[1], in[2], 1.0f}, n[4] = {in[3], in[4], in[5], 0.0f};
    float wp[4], wn[4];
    mat_vec(world, p, wp);
    mat_vec(view_proj, wp, o->pos);
    mat_vec(world_it, n, wn);
    o->attr[0] = wp[0]; o->attr[1] = wp[1]; o->attr[2] = wp[2];
    o->attr[3] = wn[0]; o->attr[4] = wn[1]; o->attr[5] = wn[2];
    o->attr[6] = in[6]; o->attr[7] = in[7];
}

static void to_window(out_vertex *o, int width, int height, int fused) {
    const float sx = (float)width / 2.0f, sy = (float)height / 2.0f, oow = 1.0f / o->pos[3];
    if (fused) {      /* the vertex stage's own viewport transform */
        o->win[0] = fmaf(o->pos[0] * oow, sx, sx); o->win[1] = fmaf(o->pos[1] * oow, sy, sy); o->win[2] = fmaf(o->pos[2] * oow, 0.5f, 0.5f);
    } else {          /* the clipper's, for the vertices it creates */
        o->win[0] = o->pos[0] * oow * sx + sx; o->win[1] = o->pos[1] * oow * sy + sy; o->win[2] = o->pos[2] * oow * 0.5f + 0.5f;
    }
    o->win[3] = oow;
}

#define MAX_LEVELS 16
typedef struct { uint8_t *px; int w, h; } mip_level;
typedef struct {
    int width, height;
    mip_level level[MAX_LEVELS]; int n_levels;      /* level 0 = the texture as uploaded; the rest as glGenerateMipmap leaves them */
    float light[3], la[3], ld[3];
    float *color; uint32_t *depth24; uint32_t *tri_id;
} target;

static int lerp8(int a, int b, int w) { return a + (((b - a) * w + 128) >> 8); }
static int wrap(int i, int n) { int r = i % n; return r < 0 ? r + n : r; }

/* Texel coordinate in 1/256 texel.  A power-of-two size scales the coordinate itself (GL_REPEAT then wraps the integer texel index);
 * any other size takes the fraction of the coordinate first (kept below 1) and scales that.  Rounded to nearest even. */
static int texel_fixed(float c, int size) {
    if ((size & (size - 1)) == 0) return (int)lrintf(c * (float)size * 256.0f);
    const float f = fminf(c - floorf(c), 0.99999994f);
    return (int)lrintf(f * (float)size * 256.0f);
}

/* One bilinear lookup in 8-bit fixed point: lerp along x, then along y.  repeat: GL_REPEAT, else clamp to the edge texels. */
static void bilinear8(const mip_level *l, float u, float v, int repeat, int out[4]) {
    const int su = texel_fixed(u, l->w) - 128, sv = texel_fixed(v, l->h) - 128;
    int i0 = su >> 8, i1 = (su >> 8) + 1, j0 = sv >> 8, j1 = (sv >> 8) + 1;
    if (repeat) { i0 = wrap(i0, l->w); i1 = wrap(i1, l->w); j0 = wrap(j0, l->h); j1 = wrap(j1, l->h); }
    else {
        i0 = i0 < 0 ? 0 : i0 > l->w - 1 ? l->w - 1 : i0; i1 = i1 < 0 ? 0 : i1 > l->w - 1 ? l->w - 1 : i1;
        j0 = j0 < 0 ? 0 : j0 > l->h - 1 ? l->h - 1 : j0; j1 = j1 < 0 ? 0 : j1 > l->h - 1 ? l->h - 1 : j1;
    }
    const int wu = su & 255, wv = sv & 255;
    for (int c = 0; c < 4; c++) {
        const int a = l->px[((size_t)j0 * l->w + i0) * 4 + c], b = l->px[((size_t)j0 * l->w + i1) * 4 + c];
        const int d = l->px[((size_t)j1 * l->w + i0) * 4 + c], e = l->px[((size_t)j1 * l->w + i1) * 4 + c];
        out[c] = lerp8(lerp8(a, b, wu), lerp8(d, e, wu), wv);
    }
}

/* glGenerateMipmap (GLUtils.cpp:215) as llvmpipe performs it: every texel of a level is ONE bilinear lookup of the level above at the
 * texel's centre, edges clamped -- for even sizes the mean of four texels with the rounding of the 8-bit lerp, not a box filter's. */
static int build_mip_chain(target *t, const uint8_t *texture, int w, int h) {
    t->level[0].px = (uint8_t *)texture; t->level[0].w = w; t->level[0].h = h; t->n_levels = 1;
    while ((w > 1 || h > 1) && t->n_levels < MAX_LEVELS) {
        const mip_level *src = &t->level[t->n_levels - 1];
        w = w > 1 ? w / 2 : 1; h = h > 1 ? h / 2 : 1;
        mip_level *dst = &t->level[t->n_levels];
        dst->w = w; dst->h = h; dst->px = malloc((size_t)w * h * 4);
        if (!dst->px) return 1;
        for (int j = 0; j < h; j++)
            for (int i = 0; i < w; i++) {
                int c[4];
                bilinear8(src, (float)((i + 0.5) / w), (float)((j + 0.5) / h), 0, c);
                for (int k = 0; k < 4; k++) dst->px[((size_t)j * w + i) * 4 + k] = (uint8_t)c[k];
            }
        t->n_levels++;
    }
    return 0;
}

/* GL_LINEAR_MIPMAP_LINEAR / GL_LINEAR (GLUtils.cpp:217-219).  duv: the texture coordinates of the fragment's 2 x 2 quad -- [0] its
 * lower left pixel (even x, even y), [1] the pixel to the right, [2] the pixel above -- which every fragment of the quad shares.
 * rho^2 = max(|d(uv)/dx|^2, |d(uv)/dy|^2) in texels; lod = log2(rho^2) / 2 with log2 read off the float's bits (exponent + mantissa
 * fraction: linear between powers of two); lod <= 0 magnifies (level 0); else levels floor(lod) and the next with the weight
 * floor(fraction * 256) in the 8-bit lerp; beyond the last level, the last level. */
static void sample_texture(const target *t, float u, float v, const float duv[3][2], float out[4]) {
    const float tw = (float)t->level[0].w, th = (float)t->level[0].h;
    const float a = (duv[1][0] - duv[0][0]) * tw, c = (duv[1][1] - duv[0][1]) * th, b = (duv[2][0] - duv[0][0]) * tw, d = (duv[2][1] - duv[0][1]) * th;
    const float rho2 = fmaxf(a * a + c * c, b * b + d * d);
    int32_t bits; memcpy(&bits, &rho2, 4);
    const float lod = (float)(bits - 0x3f800000) * 0x1p-23f * 0.5f;
    int col[4];
    const int last = t->n_levels - 1;
    if (!(lod > 0.0f)) bilinear8(&t->level[0], u, v, 1, col);
    else {
        const float fl = floorf(lod);
        if (fl >= (float)last) bilinear8(&t->level[last], u, v, 1, col);
        else {
            const int l0 = (int)fl, w8 = (int)floorf((lod - fl) * 256.0f);
            int c0[4], c1[4];
            bilinear8(&t->level[l0], u, v, 1, c0); bilinear8(&t->level[l0 + 1], u, v, 1, c1);
            for (int k = 0; k < 4; k++) col[k] = lerp8(c0[k], c1[k], w8);
        }
    }
    for (int k = 0; k < 4; k++) out[k] = (float)col[k] * (1.0f / 255.0f);
}

/* Frag_LightingSimple.frag:29-41 */
static void fragment_stage(const target *t, const float *a /* 8 varyings */, const float duv[3][2], float out[4]) {
    const float nl = 1.0f / sqrtf((a[5] * a[5] + a[4] * a[4]) + a[3] * a[3]);
    const float n[3] = {a[3] * nl, a[4] * nl, a[5] * nl};
    const float m[3] = {-t->light[0], -t->light[1], -t->light[2]};
    const float ml = 1.0f / sqrtf((m[2] * m[2] + m[1] * m[1]) + m[0] * m[0]);
    const float l[3] = {m[0] * ml, m[1] * ml, m[2] * ml};
    const float d = (n[2] * l[2] + n[1] * l[1]) + n[0] * l[0];
    const float cosa = fminf(fmaxf(d, 0.0f), 1.0f);
    float tx[4];
    sample_texture(t, a[6], a[7], duv, tx);
    for (int c = 0; c < 3; c++) out[c] = (t->la[c] + cosa * t->ld[c]) * tx[c];
    out[3] = 1.0f * tx[3];
}

static void raster_triangle(const target *t, const out_vertex *v0, const out_vertex *v1, const out_vertex *v2, uint32_t id) {
    const out_vertex *v[3] = {v0, v1, v2};
    const float *win[3] = {v0->win, v1->win, v2->win};
    int64_t fx[3], fy[3];
    for (int i = 0; i < 3; i++) {
        fx[i] = (int64_t)lrintf((win[i][0] - 0.5f) * 256.0f);
        fy[i] = (int64_t)lrintf((win[i][1] - 0.5f) * 256.0f);
    }
    (void)v;
    const int64_t area = (fx[1] - fx[0]) * (fy[2] - fy[0]) - (fx[2] - fx[0]) * (fy[1] - fy[0]);
    if (area <= 0) return;                       /* clockwise: a back face (App.cpp:50-51); zero area: nothing to draw */
    int64_t minx = fx[0], maxx = fx[0], miny = fy[0], maxy = fy[0];
    for (int i = 1; i < 3; i++) { if (fx[i] < minx) minx = fx[i]; if (fx[i] > maxx) maxx = fx[i]; if (fy[i] < miny) miny = fy[i]; if (fy[i] > maxy) maxy = fy[i]; }
    int64_t x_lo = (minx + 255) >> 8, x_hi = maxx >> 8, y_lo = (miny + 255) >> 8, y_hi = maxy >> 8;
    if (x_lo < 0) x_lo = 0;
    if (y_lo < 0) y_lo = 0;
    if (x_hi > t->width - 1) x_hi = t->width - 1;
    if (y_hi > t->height - 1) y_hi = t->height - 1;
    if (x_lo > x_hi || y_lo > y_hi) return;
    /* planes from (P, Q, R) = (v1, v0, v2) */
    const float *P = win[1], *Q = win[0], *R = win[2];
    const float dx01 = P[0] - Q[0], dy01 = P[1] - Q[1], dx20 = R[0] - P[0], dy20 = R[1] - P[1];
    const float ooa = 1.0f / (dx01 * dy20 - dx20 * dy01);
    const float dy20_ooa = dy20 * ooa, dy01_ooa = dy01 * ooa, dx01_ooa = dx01 * ooa, dx20_ooa = dx20 * ooa;
    const float xc = P[0] - 0.5f, yc = P[1] - 0.5f;
    float a0[10], dadx[10], dady[10];      /* z, 1/w, 8 varyings (premultiplied by 1/w) */
    for (int k = 0; k < 10; k++) {
        float aP, aQ, aR;
        if (k < 2) { aP = P[2 + k]; aQ = Q[2 + k]; aR = R[2 + k]; }
        else { aP = v1->attr[k - 2] * P[3]; aQ = v0->attr[k - 2] * Q[3]; aR = v2->attr[k - 2] * R[3]; }
        const float da01 = aP - aQ, da20 = aR - aP;
        dadx[k] = da01 * dy20_ooa - da20 * dy01_ooa;
        dady[k] = da20 * dx01_ooa - da01 * dx20_ooa;
        a0[k] = aP - (dadx[k] * xc + dady[k] * yc);
    }
    for (int64_t y = y_lo; y <= y_hi; y++) {
        for (int64_t x = x_lo; x <= x_hi; x++) {
            int inside = 1;
            for (int i = 0; i < 3 && inside; i++) {
                const int j = (i + 1) % 3;
                const int64_t ex = fx[j] - fx[i], ey = fy[j] - fy[i];
                const int64_t e = ex * (y * 256 - fy[i]) - ey * (x * 256 - fx[i]);
                if (e < 0 || (e == 0 && !(ey < 0 || (ey == 0 && ex > 0)))) inside = 0;
            }
            if (!inside) continue;
            const float xf = (float)x, yf = (float)y;
            float z = fmaf(dady[0], yf, fmaf(dadx[0], xf, a0[0]));
            z = fminf(fmaxf(z, 0.0f), 1.0f);
            const uint32_t d24 = (uint32_t)lrintf(z * 16777215.0f);
            const size_t pi = (size_t)y * t->width + (size_t)x;
            if (!(d24 <= t->depth24[pi])) continue;
            const float w = 1.0f / fmaf(dady[1], yf, fmaf(dadx[1], xf, a0[1]));
            float a[8], col[4], duv[3][2];
            for (int k = 0; k < 8; k++) a[k] = fmaf(dady[2 + k], yf, fmaf(dadx[2 + k], xf, a0[2 + k])) * w;
            /* the texture coordinates at three pixels of the fragment's quad (covered or not: the planes extend beyond the triangle) */
            for (int q = 0; q < 3; q++) {
                const float qx = (float)((x & ~(int64_t)1) + (q == 1)), qy = (float)((y & ~(int64_t)1) + (q == 2));
                const float qw = 1.0f / fmaf(dady[1], qy, fmaf(dadx[1], qx, a0[1]));
                duv[q][0] = fmaf(dady[8], qy, fmaf(dadx[8], qx, a0[8])) * qw; duv[q][1] = fmaf(dady[9], qy, fmaf(dadx[9], qx, a0[9])) * qw;
            }
            fragment_stage(t, a, duv, col);
            memcpy(t->color + 4 * pi, col, sizeof col);
            t->depth24[pi] = d24;
            if (t->tri_id) t->tri_id[pi] = id;
        }
    }
}

/* A new vertex on a polygon edge, interpolated from its end `a` (the one nearer to the plane): a + t * (b - a) for the clip position and every output. */
static void clip_vertex(const out_vertex *a, const out_vertex *b, float t, out_vertex *o, int width, int height) {
    for (int k = 0; k < 4; k++) o->pos[k] = a->pos[k] + t * (b->pos[k] - a->pos[k]);
    for (int k = 0; k < 8; k++) o->attr[k] = a->attr[k] + t * (b->attr[k] - a->attr[k]);
    to_window(o, width, height, 0);
}
static float plane_distance(const out_vertex *v, int plane) {      /* >= 0 inside; planes in the order they are applied */
    switch (plane) {
    case 0: return v->pos[3] - v->pos[0];
    case 1: return v->pos[3] + v->pos[0];
    case 2: return v->pos[3] - v->pos[1];
    case 3: return v->pos[3] + v->pos[1];
    case 4: return v->pos[3] + v->pos[2];
    default: return v->pos[3] - v->pos[2];
    }
}
static int clip_plane(const out_vertex *in, int n, int plane, out_vertex *out, int width, int height) {
    int m = 0;
    const out_vertex *prev = &in[0];
    float dp_prev = plane_distance(prev, plane);
    for (int i = 1; i <= n; i++) {
        const out_vertex *cur = &in[i % n];
        const float dp = plane_distance(cur, plane);
        if (dp_prev >= 0.0f) out[m++] = *prev;
        if ((dp < 0.0f) != (dp_prev < 0.0f)) {
            if (fabsf(dp_prev) < fabsf(dp)) clip_vertex(prev, cur, dp_prev / (dp_prev - dp), &out[m++], width, height);
            else clip_vertex(cur, prev, dp / (dp - dp_prev), &out[m++], width, height);
        }
        prev = cur; dp_prev = dp;
    }
    return m;
}

/* words: viewProj[16], light_dir[3], La[3], Ld[3], 1 pad, then world[16] + worldIT[16] per draw.  color [h][w][4] and depth24 [h][w]
 * are read and updated in place (row 0 = bottom, as GL reads them back); tri_id (optional) receives draw * n_triangles + triangle
 * for every pixel a fragment wrote. */
int raster_oracle_draw(const float *vertices, size_t n_vert, const uint32_t *indices, size_t n_idx, const float *words, size_t n_draws,
                       const uint8_t *texture, int tex_w, int tex_h, float *color, uint32_t *depth24, uint32_t *tri_id, int width, int height) {
    target t;
    t.width = width; t.height = height;
    if (build_mip_chain(&t, texture, tex_w, tex_h)) return 2;
    memcpy(t.light, words + 16, 12); memcpy(t.la, words + 19, 12); memcpy(t.ld, words + 22, 12);
    t.color = color; t.depth24 = depth24; t.tri_id = tri_id;
    for (size_t i = 0; i < n_idx; i++) if (indices[i] >= n_vert) return 1;
    out_vertex *vs = malloc(n_vert * sizeof *vs);
    if (!vs) return 2;
    for (size_t d = 0; d < n_draws; d++) {
        const float *world = words + 26 + 32 * d, *world_it = world + 16;
        for (size_t i = 0; i < n_vert; i++) { vertex_stage(vertices + 8 * i, words, world, world_it, &vs[i]); to_window(&vs[i], width, height, 1); }
        for (size_t k = 0; k + 2 < n_idx; k += 3) {
            const out_vertex *v[3] = {&vs[indices[k]], &vs[indices[k + 1]], &vs[indices[k + 2]]};
            const uint32_t id = (uint32_t)(d * (n_idx / 3) + k / 3);
            unsigned any_out = 0, all_out = 63;
            for (int q = 0; q < 3; q++) {
                unsigned mask = 0;
                for (int pl = 0; pl < 6; pl++) if (!(plane_distance(v[q], pl) >= 0.0f)) mask |= 1u << pl;
                any_out |= mask; all_out &= mask;
            }
            if (!any_out) { raster_triangle(&t, v[0], v[1], v[2], id); continue; }
            if (all_out) continue;
            out_vertex p0[12], p1[12], *src = p0, *dst = p1;
            p0[0] = *v[0]; p0[1] = *v[1]; p0[2] = *v[2];
            int n = 3;
            for (int pl = 0; pl < 6 && n >= 3; pl++) if (any_out & (1u << pl)) {
                n = clip_plane(src, n, pl, dst, width, height);
                out_vertex *sw = src; src = dst; dst = sw;
            }
            /* the polygon as a fan that keeps its first vertex last in every triangle; a clipped triangle's pieces are culled by the sign
               of their area in FLOAT window coordinates, e x f with e = v0 - v2, f = v1 - v2 (zero passes), before the rasteriser applies
               its own test in snapped coordinates */
            for (int q = 2; q < n; q++) {
                const float *a = src[q - 1].win, *b = src[q].win, *c = src[0].win;
                const float ex = a[0] - c[0], ey = a[1] - c[1], fx = b[0] - c[0], fy = b[1] - c[1];
                if (ex * fy - ey * fx < 0.0f) continue;
                raster_triangle(&t, &src[q - 1], &src[q], &src[0], id);
            }
        }
    }
    free(vs);
    for (int l = 1; l < t.n_levels; l++) free(t.level[l].px);
    return 0;
}
