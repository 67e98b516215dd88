/* Feasibility count, CPU only: node visits and triangle tests per ray for (a) the octree as the shader walks it with the children
 * nearest first and (b) the same tree with every triangle kept once (in the leaf of its centroid) and tight boxes per node.
 * Built and run by tools/experiments/bvh_sim.py. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

typedef struct { float x, y, z; } v3;
typedef struct { v3 lo, hi; } box;
typedef struct { uint32_t word, tri_start, tri_count, child[8], mask; box cell, tight; uint32_t u_start, u_count; } node;

static node *N; static uint32_t n_nodes, cap;
static const uint32_t *W; static const float *V; static const uint32_t *R;

static uint32_t add_node(uint32_t word, box cell) {
    if (n_nodes == cap) { cap = cap ? cap * 2 : 1 << 20; N = realloc(N, (size_t)cap * sizeof(node)); }
    uint32_t id = n_nodes++;
    node *n = &N[id]; memset(n, 0, sizeof *n);
    uint32_t info = W[word];
    n->word = word; n->tri_start = W[word + 1]; n->tri_count = info >> 16; n->mask = (info >> 8) & 0xff; n->cell = cell;
    n->tight.lo = (v3){INFINITY, INFINITY, INFINITY}; n->tight.hi = (v3){-INFINITY, -INFINITY, -INFINITY};
    v3 mid = {(cell.hi.x + cell.lo.x) / 2.0f, (cell.hi.y + cell.lo.y) / 2.0f, (cell.hi.z + cell.lo.z) / 2.0f};
    uint32_t ci = 0; const uint32_t mask = n->mask;
    for (uint32_t i = 0; i < 8; i++) if (mask & (1u << i)) {
        box cb = {{(i & 1) ? mid.x : cell.lo.x, (i & 2) ? mid.y : cell.lo.y, (i & 4) ? mid.z : cell.lo.z}, {(i & 1) ? cell.hi.x : mid.x, (i & 2) ? cell.hi.y : mid.y, (i & 4) ? cell.hi.z : mid.z}};
        uint32_t c = add_node(W[word + 2 + ci], cb); ci++;
        N[id].child[i] = c;
    }
    return id;
}
static v3 vert(uint32_t i) { return (v3){V[4 * i], V[4 * i + 1], V[4 * i + 2]}; }
static void grow(box *b, v3 p) { b->lo.x = fminf(b->lo.x, p.x); b->lo.y = fminf(b->lo.y, p.y); b->lo.z = fminf(b->lo.z, p.z); b->hi.x = fmaxf(b->hi.x, p.x); b->hi.y = fmaxf(b->hi.y, p.y); b->hi.z = fmaxf(b->hi.z, p.z); }

static uint32_t *u_of_ref;      /* unique triangle id per reference */
static uint32_t *u_first_ref;   /* one reference per unique triangle */
static uint32_t n_unique;
static uint32_t *u_list;        /* unique triangles ordered by owning node */

typedef struct { uint32_t a, b, c, m, r; } key_t_;
static int cmpk(const void *x, const void *y) { const key_t_ *p = x, *q = y; if (p->a != q->a) return p->a < q->a ? -1 : 1; if (p->b != q->b) return p->b < q->b ? -1 : 1; if (p->c != q->c) return p->c < q->c ? -1 : 1; if (p->m != q->m) return p->m < q->m ? -1 : 1; return p->r < q->r ? -1 : (p->r > q->r); }

static float tri_t(v3 o, v3 d, uint32_t ref) {
    v3 v1 = vert(R[4 * ref]), v2 = vert(R[4 * ref + 1]), v3_ = vert(R[4 * ref + 2]);
    v3 e1 = {v2.x - v1.x, v2.y - v1.y, v2.z - v1.z}, e2 = {v3_.x - v1.x, v3_.y - v1.y, v3_.z - v1.z};
    v3 p = {d.y * e2.z - d.z * e2.y, d.z * e2.x - d.x * e2.z, d.x * e2.y - d.y * e2.x};
    float det = e1.x * p.x + e1.y * p.y + e1.z * p.z;
    if (fabsf(det) < 1e-15f) return -1;
    float inv = 1.0f / det; v3 t = {o.x - v1.x, o.y - v1.y, o.z - v1.z};
    float u = inv * (t.x * p.x + t.y * p.y + t.z * p.z); if (u < 0 || u > 1) return -1;
    v3 q = {t.y * e1.z - t.z * e1.y, t.z * e1.x - t.x * e1.z, t.x * e1.y - t.y * e1.x};
    float v = inv * (d.x * q.x + d.y * q.y + d.z * q.z); if (v < 0 || u + v > 1) return -1;
    return (e2.x * q.x + e2.y * q.y + e2.z * q.z) * inv;
}
static int slab(v3 o, v3 inv, box b, float closest, float *tmin_out) {
    float t1 = (b.lo.x - o.x) * inv.x, t2 = (b.hi.x - o.x) * inv.x, t3 = (b.lo.y - o.y) * inv.y, t4 = (b.hi.y - o.y) * inv.y, t5 = (b.lo.z - o.z) * inv.z, t6 = (b.hi.z - o.z) * inv.z;
    float tmin = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6)), tmax = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
    *tmin_out = tmin;
    return tmax > 0 && tmin < tmax && tmin <= closest;
}

/* out: [0] pops, [1] tri tests, [2] box tests, [3] t bits */
static float walk(v3 o, v3 d, int tight, int sort_by_tmin, uint64_t *cnt) {
    v3 inv = {1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
    uint32_t key = ((d.x < 0) | ((d.y < 0) << 1) | ((d.z < 0) << 2));   /* nearest octant = key */
    float closest = INFINITY, tm;
    static uint32_t stack[4096]; int sp = 0;
    if (!slab(o, inv, tight ? N[0].tight : N[0].cell, closest, &tm)) return closest;
    stack[sp++] = 0;
    while (sp) {
        node *n = &N[stack[--sp]];
        cnt[0]++;
        if (tight) { for (uint32_t i = 0; i < n->u_count; i++) { cnt[1]++; float t = tri_t(o, d, u_first_ref[u_list[n->u_start + i]]); if (t >= 0 && t < closest) closest = t; } }
        else { for (uint32_t i = 0; i < n->tri_count; i++) { cnt[1]++; float t = tri_t(o, d, n->tri_start + i); if (t >= 0 && t < closest) closest = t; } }
        uint32_t ord[8]; float tms[8]; int k = 0;
        /* push far first so that near pops first */
        for (int j = 7; j >= 0; j--) {
            uint32_t oct = (uint32_t)j ^ key;   /* j = 0 is nearest */
            if (!(n->mask & (1u << oct))) continue;
            node *c = &N[n->child[oct]];
            if (tight && c->tight.lo.x > c->tight.hi.x) continue;      /* empty subtree */
            cnt[2]++;
            if (slab(o, inv, tight ? c->tight : c->cell, closest, &tm)) { ord[k] = n->child[oct]; tms[k] = tm; k++; }
        }
        if (sort_by_tmin) for (int a = 1; a < k; a++) { uint32_t oo = ord[a]; float tt = tms[a]; int b = a - 1; while (b >= 0 && tms[b] < tt) { ord[b + 1] = ord[b]; tms[b + 1] = tms[b]; b--; } ord[b + 1] = oo; tms[b + 1] = tt; }
        for (int a = 0; a < k; a++) stack[sp++] = ord[a];
    }
    return closest;
}

int bvh_sim(const float *verts, const uint32_t *refs, uint32_t n_ref, const uint32_t *words, const float *bmin, const float *bmax,
            const float *rays, uint32_t n_rays, uint64_t *out /* 3 modes x 3 counters + mismatches + n_unique + n_nodes + nonempty */) {
    V = verts; R = refs; W = words; n_nodes = 0;
    box root = {{bmin[0], bmin[1], bmin[2]}, {bmax[0], bmax[1], bmax[2]}};
    add_node(0, root);
    /* unique triangles */
    key_t_ *k = malloc((size_t)n_ref * sizeof *k);
    for (uint32_t i = 0; i < n_ref; i++) k[i] = (key_t_){R[4 * i], R[4 * i + 1], R[4 * i + 2], R[4 * i + 3], i};
    qsort(k, n_ref, sizeof *k, cmpk);
    u_of_ref = malloc((size_t)n_ref * 4); u_first_ref = malloc((size_t)n_ref * 4); n_unique = 0;
    for (uint32_t i = 0; i < n_ref; i++) {
        if (i == 0 || k[i].a != k[i - 1].a || k[i].b != k[i - 1].b || k[i].c != k[i - 1].c || k[i].m != k[i - 1].m) u_first_ref[n_unique++] = k[i].r;
        u_of_ref[k[i].r] = n_unique - 1;
    }
    free(k);
    /* owner: the leaf of the centroid */
    uint32_t *owner = malloc((size_t)n_unique * 4), *count = calloc(n_nodes + 1, 4);
    for (uint32_t u = 0; u < n_unique; u++) {
        uint32_t r = u_first_ref[u]; v3 a = vert(R[4 * r]), b = vert(R[4 * r + 1]), c = vert(R[4 * r + 2]);
        v3 g = {(a.x + b.x + c.x) / 3, (a.y + b.y + c.y) / 3, (a.z + b.z + c.z) / 3};
        uint32_t ni = 0;
        for (;;) {
            node *n = &N[ni]; if (!n->mask) break;
            v3 mid = {(n->cell.hi.x + n->cell.lo.x) / 2.0f, (n->cell.hi.y + n->cell.lo.y) / 2.0f, (n->cell.hi.z + n->cell.lo.z) / 2.0f};
            uint32_t o = (g.x >= mid.x) | ((g.y >= mid.y) << 1) | ((g.z >= mid.z) << 2);
            if (!(n->mask & (1u << o))) break;      /* keep it at this inner node */
            ni = n->child[o];
        }
        owner[u] = ni; count[ni]++;
    }
    uint32_t acc = 0; for (uint32_t i = 0; i < n_nodes; i++) { N[i].u_start = acc; N[i].u_count = count[i]; acc += count[i]; count[i] = 0; }
    u_list = malloc((size_t)n_unique * 4);
    for (uint32_t u = 0; u < n_unique; u++) { node *n = &N[owner[u]]; u_list[n->u_start + count[owner[u]]++] = u; }
    /* tight boxes: children have larger indices than their parents (pre-order) */
    uint64_t nonempty = 0;
    for (int64_t i = n_nodes - 1; i >= 0; i--) {
        node *n = &N[i];
        for (uint32_t j = 0; j < n->u_count; j++) { uint32_t r = u_first_ref[u_list[n->u_start + j]]; for (int q = 0; q < 3; q++) grow(&n->tight, vert(R[4 * r + q])); }
        for (int o = 0; o < 8; o++) if (n->mask & (1u << o)) { node *c = &N[n->child[o]]; if (c->tight.lo.x <= c->tight.hi.x) { grow(&n->tight, c->tight.lo); grow(&n->tight, c->tight.hi); } }
        if (n->tight.lo.x <= n->tight.hi.x) nonempty++;
    }
    memset(out, 0, 16 * 8);
    for (uint32_t i = 0; i < n_rays; i++) {
        v3 o = {rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]}, d = {rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]};
        float t0 = walk(o, d, 0, 0, out + 0), t1 = walk(o, d, 1, 0, out + 3), t2 = walk(o, d, 1, 1, out + 6);
        if (memcmp(&t0, &t1, 4) || memcmp(&t0, &t2, 4)) out[9]++;
        if (!isinf(t0)) out[13]++;
    }
    out[10] = n_unique; out[11] = n_nodes; out[12] = nonempty;
    return 0;
}
