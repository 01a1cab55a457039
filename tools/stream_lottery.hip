// Does the placement lottery of the fused FDTD kernel show in a plain 18-read + 6-write plane march, and
// does it go away when the arrays advance through memory at different rates?
//   hipcc --offload-arch=gfx950 -O3 -o stream_lottery tools/stream_lottery.hip && ./stream_lottery [rounds]
// Per round: ONE device allocation holding 24 arrays of 1024 x 1025 x 1026 fp32 (as the engine's arena does),
// then the same march (16 rows x 64 lanes x 16 B per workgroup, 24 planes per workgroup, tiles dealt in runs of
// 4 per XCD) timed for several layouts inside that allocation:
//   same     every array the same plane pitch (the engine's layout)
//   skew N   array q's plane pitch is longer by q x N bytes, so the distance between the places two arrays
//            are streamed from changes from plane to plane
//   perm     same pitch, but array q keeps plane k in slot (k * (q + 1)) % 1031: no extra memory per plane
// The allocation is freed and made again for the next round.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Lay { float4* p[24]; unsigned long long sz4[24]; int mul[24]; int mod; };    // base, plane pitch (in float4) and plane order per array: plane k sits in slot (k * mul) % mod

__device__ __forceinline__ unsigned xcd_grouped(unsigned bid, unsigned nblk, unsigned G) {
    const unsigned span = 8u * G, bulk = nblk / span * span;
    if (bid >= bulk) return bid;
    const unsigned x = bid & 7u, pos = bid >> 3;
    return ((pos / G) * 8u + x) * G + pos % G;
}

template <int R, int W>
__global__ void __launch_bounds__(1024) k_march(Lay a, int px4, int ny, int nz, int chunk, int nbx, int nby) {
    const int lane = threadIdx.x & 63, wy = threadIdx.x >> 6;
    unsigned bid = xcd_grouped(blockIdx.x, gridDim.x, 4);
    const int bx = bid % nbx; bid /= nbx;
    const int by = bid % nby, bz = bid / nby;
    const int j = by * 16 + wy; if (j >= ny) return;
    const int k0 = bz * chunk, k1 = min(k0 + chunk, nz);
    const size_t row = (size_t)j * px4 + bx * 64 + lane;
    for (int k = k0; k < k1; ++k) {
        float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < R; ++r) { float4 v = a.p[r][(size_t)((k * a.mul[r]) % a.mod) * a.sz4[r] + row]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
#pragma unroll
        for (int w = 0; w < W; ++w) { v4f t = {acc.x + w, acc.y, acc.z, acc.w}; __builtin_nontemporal_store(t, (v4f*)&a.p[R + w][(size_t)((k * a.mul[R + w]) % a.mod) * a.sz4[R + w] + row]); }
    }
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 6;
    const int px = 1024, ny = 1025, nz = argc > 2 ? atoi(argv[2]) : 1026, px4 = px / 4;
    const size_t plane4 = (size_t)px4 * ny;
    // skew: array q's plane pitch grows by (q % group) * skew bytes; pad: every array's plane pitch grows by pad bytes;
    // perm: array q keeps plane k in slot (k * (q + 1)) % 1031 (a prime above the plane count) instead of slot k
    // skew: array q's plane pitch grows by (q % group) * skew bytes; pad: every array's plane pitch grows by pad bytes;
    // perm: array q keeps plane k in slot (k * (q + 1)) % 1031; quad: equal pitches, but array q starts where it would
    // start in the skewed layout (q (q - 1) / 2 * quad * nz bytes further on); gap: equal pitches, `gap` bytes between arrays
    // quad: equal pitches, array q starts q (q - 1) / 2 * quad * nz bytes further on than with uniform spacing;
    // rnd: equal pitches, a pseudo-random gap below `rnd` bytes (4 KB granules, seed `seed`) in front of every array
    struct V { const char* name; size_t skew; int group; size_t pad; int perm; size_t quad; size_t gap; size_t rnd; unsigned seed; };
    const V vs[] = {{"same", 0, 24, 0, 0, 0, 0, 0, 0}, {"quad 516K", 0, 24, 0, 0, 528384, 0, 0, 0},
                    {"gap 1G", 0, 24, 0, 0, 0, (size_t)1 << 30, 0, 0}, {"gap 2G", 0, 24, 0, 0, 0, (size_t)2 << 30, 0, 0}, {"gap 3G", 0, 24, 0, 0, 0, (size_t)3 << 30, 0, 0},
                    {"gap 4G", 0, 24, 0, 0, 0, (size_t)4 << 30, 0, 0}, {"gap 6G", 0, 24, 0, 0, 0, (size_t)6 << 30, 0, 0}, {"gap 8G", 0, 24, 0, 0, 0, (size_t)8 << 30, 0, 0},
                    {"gap 9.3G", 0, 24, 0, 0, 0, (size_t)9523 << 20, 0, 0}, {"same", 0, 24, 0, 0, 0, 0, 0, 0}};
    const int nv = sizeof vs / sizeof vs[0];
    const size_t room = 24 * (plane4 * 16 * (size_t)nz + 4096) + std::max((size_t)276 * 528384 * nz, (size_t)24 * ((size_t)9523 << 20)) + (1u << 20);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("rounds %d, one allocation of %.1f GB per round; ms per march (best of 3), GB/s for the 24 x 4.3 GB moved\n", rounds, room / 1e9);
    printf("%-6s", "round"); for (int v = 0; v < nv; ++v) printf(" %13s |", vs[v].name); printf("\n");
    for (int r = 0; r < rounds; ++r) {
        char* base = nullptr;
        CK(hipMalloc((void**)&base, room));
        CK(hipMemset(base, 0, room));
        printf("%-6d", r);
        for (int v = 0; v < nv; ++v) {
            Lay a; size_t off = 0;
            a.mod = vs[v].perm ? 1031 : 1 << 30;
            unsigned long long rs = 0x9E3779B97F4A7C15ull * (vs[v].seed + 1);
            for (int q = 0; q < 24; ++q) {
                const size_t sz_bytes = plane4 * 16 + (size_t)(q % vs[v].group) * vs[v].skew + vs[v].pad;
                if (vs[v].rnd) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; off += (size_t)(rs % (vs[v].rnd >> 12)) << 12; }
                a.p[q] = (float4*)(base + off + (size_t)q * (q - 1) / 2 * vs[v].quad * nz);
                a.sz4[q] = sz_bytes / 16;
                a.mul[q] = vs[v].perm ? q + 1 : 1;
                off += sz_bytes * (vs[v].perm ? 1031 : nz) + 4096 + vs[v].gap;
            }
            if (off + (size_t)276 * vs[v].quad * nz > room) { printf(" layout too big\n"); return 1; }
            const int nbx = px4 / 64, nby = (ny + 15) / 16, nbz = (nz + 23) / 24;
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL((k_march<18, 6>), dim3((unsigned)nbx * nby * nbz), dim3(1024), 0, 0, a, px4, ny, nz, 24, nbx, nby);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            CK(hipGetLastError());
            printf(" %6.2f %5.0f |", best, (double)plane4 * 16 * nz * 24 / best / 1e6);
        }
        printf("\n"); fflush(stdout);
        CK(hipFree(base));
    }
    return 0;
}
