// Load-path experiment for the traversal kernel: chains of dependent gathers of 64-byte records from a table far larger than the
// caches, by a persistent grid shaped like traverse_kernel's (5 workgroups of 256 threads per CU), with only some lanes of every
// wave holding a chain (the traversal runs at about 26 of 64).  Three ways to fetch a lane's record:
//   0  the lane loads its own record with four 16-byte loads (what the kernels do): every load instruction touches one cache
//      line per active lane
//   1  four lanes share a record, one 16-byte quarter each, written straight to LDS (global_load_lds_dwordx4), owners compacted
//      by rank so that an instruction serves 16 owners: one line per four lanes; the owner then reads its 64 bytes from LDS
//   2  the same without compaction: round r serves the owners with lane % 4 == r (address broadcast inside the quad)
// Prints ns per step of a wave and records per microsecond per CU.
// Build: hipcc -O3 --offload-arch=gfx950 -o gather_bench gather_bench.hip ; run: ./gather_bench [active lanes] [records in millions]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct alignas(64) Rec { uint4 q[4]; };

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *glb_ptr_t;

template <int MODE>
__global__ __launch_bounds__(256) void chase(const Rec *__restrict__ tab, uint32_t n_rec, uint32_t steps, uint32_t n_active, uint32_t *__restrict__ out) {
    __shared__ uint4 stage[4][64][4];      // per wave: 64 records of 64 bytes
    __shared__ uint32_t idx[4][64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    // a fixed set of n_active lanes per wave, scattered
    const bool active = (mix(lane * 2654435761u + (gid >> 6) * 40503u) % 64u) < n_active;
    uint32_t cur = mix(gid) % n_rec;
    uint32_t acc = 0;
    const unsigned long long m = __ballot(active);
    const uint32_t n = (uint32_t)__popcll(m);
    const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    for (uint32_t s = 0; s < steps; s++) {
        uint4 a, b, c, d;
        if (MODE == 0) {
            if (active) {
                const uint4 *p = tab[cur].q;
                a = p[0]; b = p[1]; c = p[2]; d = p[3];
            }
        } else if (MODE == 1) {
            if (active) idx[wave][rank] = cur;
            __builtin_amdgcn_wave_barrier();
            for (uint32_t r = 0; r * 16u < n; r++) {
                const uint32_t k = r * 16u + (lane >> 2);
                if (k < n) {
                    const uint32_t ni = idx[wave][k];
                    const char *src = reinterpret_cast<const char *>(tab + ni) + (lane & 3u) * 16u;
                    __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)&stage[wave][r * 16u][0], 16, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (active) { a = stage[wave][rank][0]; b = stage[wave][rank][1]; c = stage[wave][rank][2]; d = stage[wave][rank][3]; }
        } else {
#pragma unroll
            for (uint32_t r = 0; r < 4; r++) {
                const uint32_t owner = (lane & ~3u) | r;
                const uint32_t ni = __shfl(cur, owner, 64);
                const bool on = (m >> owner) & 1ull;
                if (on) {
                    const char *src = reinterpret_cast<const char *>(tab + ni) + (lane & 3u) * 16u;
                    __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)&stage[wave][r * 16u][0], 16, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const uint32_t at = (lane & 3u) * 16u + (lane >> 2);
            if (active) { a = stage[wave][at][0]; b = stage[wave][at][1]; c = stage[wave][at][2]; d = stage[wave][at][3]; }
        }
        if (active) {
            const uint32_t v = a.x ^ b.y ^ c.z ^ d.w;
            acc += v;
            cur = mix(v ^ s) % n_rec;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (active) out[gid] = acc;
}


// Record-size experiment: per-lane loads of Q x 16 bytes from records of Q*16 bytes, aligned to their size.
template <int Q>
__global__ __launch_bounds__(256) void chase_size(const uint4 *__restrict__ tab, uint32_t n_rec, uint32_t steps, uint32_t n_active, uint32_t *__restrict__ out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = (mix(lane * 2654435761u + (gid >> 6) * 40503u) % 64u) < n_active;
    uint32_t cur = mix(gid) % n_rec;
    uint32_t acc = 0;
    for (uint32_t s = 0; s < steps; s++) {
        if (active) {
            const uint4 *p = tab + (size_t)cur * Q;
            uint4 v[Q];
#pragma unroll
            for (int q = 0; q < Q; q++) v[q] = p[q];
            uint32_t x = 0;
#pragma unroll
            for (int q = 0; q < Q; q++) x ^= v[q].x ^ v[q].w;
            acc += x;
            cur = mix(x ^ s) % n_rec;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (active) out[gid] = acc;
}

int main(int argc, char **argv) {
    const uint32_t n_active = argc > 1 ? (uint32_t)atoi(argv[1]) : 26u;
    const uint32_t n_rec = (uint32_t)((argc > 2 ? atof(argv[2]) : 5.0) * 1000000.0);
    const uint32_t steps = 400;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const uint32_t n_cu = (uint32_t)prop.multiProcessorCount;
    const uint32_t grid = n_cu * 5u;
    Rec *tab; uint32_t *out;
    CHECK(hipMalloc(&tab, (size_t)n_rec * sizeof(Rec)));
    CHECK(hipMalloc(&out, (size_t)grid * 256u * 4u));
    {
        std::vector<Rec> h(n_rec);
        uint32_t x = 12345u;
        for (uint32_t i = 0; i < n_rec; i++) for (int q = 0; q < 4; q++) {
            x = x * 1664525u + 1013904223u; h[i].q[q].x = x; x = x * 1664525u + 1013904223u; h[i].q[q].y = x;
            x = x * 1664525u + 1013904223u; h[i].q[q].z = x; x = x * 1664525u + 1013904223u; h[i].q[q].w = x;
        }
        CHECK(hipMemcpy(tab, h.data(), (size_t)n_rec * sizeof(Rec), hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<uint32_t> ref;
    for (int mode = 0; mode < 3; mode++) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
            CHECK(hipMemset(out, 0, (size_t)grid * 256u * 4u));
            CHECK(hipEventRecord(e0));
            if (mode == 0) chase<0><<<grid, 256>>>(tab, n_rec, steps, n_active, out);
            else if (mode == 1) chase<1><<<grid, 256>>>(tab, n_rec, steps, n_active, out);
            else chase<2><<<grid, 256>>>(tab, n_rec, steps, n_active, out);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        std::vector<uint32_t> h((size_t)grid * 256u);
        CHECK(hipMemcpy(h.data(), out, h.size() * 4u, hipMemcpyDeviceToHost));
        bool same = true;
        if (mode == 0) ref = h; else same = ref == h;
        const double recs = (double)grid * 4.0 * n_active * steps;
        printf("mode %d  active %u  table %.0f MB  %.3f ms  %.1f ns/step  %.1f records/us/CU  %s\n", mode, n_active, n_rec * 64.0 / 1e6, best,
               best * 1e6 / steps, recs / (best * 1e3) / n_cu, same ? "same" : "DIFFERENT");
    }
    for (int q = 1; q <= 12; q = q == 1 ? 2 : q == 2 ? 3 : q == 3 ? 4 : q == 4 ? 6 : q == 6 ? 8 : 13) {
        const uint32_t nr = (uint32_t)((size_t)n_rec * 4u / q);
        float best = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
            CHECK(hipEventRecord(e0));
            const uint4 *t4 = reinterpret_cast<const uint4 *>(tab);
            if (q == 1) chase_size<1><<<grid, 256>>>(t4, nr, steps, n_active, out);
            else if (q == 2) chase_size<2><<<grid, 256>>>(t4, nr, steps, n_active, out);
            else if (q == 3) chase_size<3><<<grid, 256>>>(t4, nr, steps, n_active, out);
            else if (q == 4) chase_size<4><<<grid, 256>>>(t4, nr, steps, n_active, out);
            else if (q == 6) chase_size<6><<<grid, 256>>>(t4, nr, steps, n_active, out);
            else chase_size<8><<<grid, 256>>>(t4, nr, steps, n_active, out);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        const double recs = (double)grid * 4.0 * n_active * steps;
        printf("record %3d B  active %u  table %.1f MB  %.3f ms  %.1f ns/step  %.1f records/us/CU\n", q * 16, n_active, n_rec * 64.0 / 1e6, best,
               best * 1e6 / steps, recs / (best * 1e3) / n_cu);
    }
    return 0;
}
