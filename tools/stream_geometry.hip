// Streaming rate of the fused kernel's array mix (18 read + 6 written arrays) for different workgroup geometries.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Ptrs { float4* p[24]; };

__device__ __forceinline__ unsigned xcd_contiguous(unsigned bid, unsigned nblk) {
    const unsigned q = nblk >> 3, r = nblk & 7u, x = bid & 7u, pos = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + pos;
}

__device__ __forceinline__ unsigned xcd_grouped(unsigned bid, unsigned nblk, unsigned G) {
    const unsigned span = 8u * G, bulk = nblk / span * span;
    if (bid >= bulk) return bid;
    const unsigned x = bid & 7u, pos = bid >> 3;
    return ((pos / G) * 8u + x) * G + pos % G;
}

// block = WX x-tiles (64 lanes each) times WY rows; marches chunk planes
template <int R, int W>
__global__ void __launch_bounds__(1024) k_march(Ptrs a, int px4, int ny, int nz, int chunk, int WX, int WY, int nbx, int nby, int nbz, int remap, int order) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wx = w % WX, wy = w / WX;
    unsigned bid = blockIdx.x;
    if (remap == 1) bid = xcd_contiguous(bid, gridDim.x); else if (remap > 1) bid = xcd_grouped(bid, gridDim.x, remap);
    int bx, by, bz;
    if (order == 0) { bx = bid % nbx; bid /= nbx; by = bid % nby; bz = bid / nby; }
    else { bz = bid % nbz; bid /= nbz; bx = bid % nbx; by = bid / nbx; }       // z fastest: consecutive blocks = stacked chunks
    const int j = by * WY + wy; if (j >= ny) return;
    const int k0 = bz * chunk, k1 = min(k0 + chunk, nz);
    size_t idx = ((size_t)k0 * ny + j) * px4 + (bx * WX + wx) * 64 + lane;
    const size_t sz = (size_t)ny * px4;
    for (int k = k0; k < k1; ++k, idx += sz) {
        float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < R; ++r) { float4 v = a.p[r][idx]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
#pragma unroll
        for (int wv = 0; wv < W; ++wv) { v4f t = {acc.x + wv, acc.y, acc.z, acc.w}; __builtin_nontemporal_store(t, (v4f*)&a.p[R + wv][idx]); }
    }
}
template <int U>
__global__ void __launch_bounds__(256) k_copy(const float4* __restrict__ s, float4* __restrict__ d, size_t n4, int nt) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = s[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (nt) { v4f t = {v[u].x, v[u].y, v[u].z, v[u].w}; __builtin_nontemporal_store(t, (v4f*)&d[i + u * stride]); } else d[i + u * stride] = v[u]; }
    }
    for (; i < n4; i += stride) d[i] = s[i];
}
// contiguous per-block slab copy: block b copies [b*len, (b+1)*len)
__global__ void __launch_bounds__(256) k_copy_slab(const float4* __restrict__ s, float4* __restrict__ d, size_t n4, size_t len) {
    size_t lo = (size_t)blockIdx.x * len, hi = lo + len < n4 ? lo + len : n4;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) d[i] = s[i];
}
int main() {
    const int px = 1024, ny = 1025, nz = 1026; const int px4 = px / 4;
    const size_t n4 = (size_t)px4 * ny * nz;
    Ptrs a;
    for (int i = 0; i < 24; ++i) { CK(hipMalloc((void**)&a.p[i], n4 * 16 + 4096)); CK(hipMemset(a.p[i], 0, n4 * 16)); }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto launch) { float best = 1e30f; for (int rep = 0; rep < 4; ++rep) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; } return best; };
    const double cb = (double)n4 * 32;
    { float ms = timeit([&] { hipMemcpyAsync(a.p[1], a.p[0], n4 * 16, hipMemcpyDeviceToDevice, 0); }); printf("hipMemcpy D2D              %.2f ms %.0f GB/s\n", ms, cb / ms / 1e6); }
    for (int g : {2048, 4096, 8192, 16384}) for (int nt : {0, 1}) {
        float ms = timeit([&] { hipLaunchKernelGGL(k_copy<4>, dim3(g), dim3(256), 0, 0, a.p[0], a.p[1], n4, nt); });
        printf("copy unroll4 grid %5d nt%d  %.2f ms %.0f GB/s\n", g, nt, ms, cb / ms / 1e6);
    }
    { float ms = timeit([&] { hipLaunchKernelGGL(k_copy<8>, dim3(2048), dim3(256), 0, 0, a.p[0], a.p[1], n4, 1); }); printf("copy unroll8 grid 2048 nt1   %.2f ms %.0f GB/s\n", ms, cb / ms / 1e6); }
    { float ms = timeit([&] { hipLaunchKernelGGL(k_copy<1>, dim3(2048), dim3(256), 0, 0, a.p[0], a.p[1], n4, 0); }); printf("copy unroll1 grid 2048       %.2f ms %.0f GB/s\n", ms, cb / ms / 1e6); }
    { size_t len = 4096; float ms = timeit([&] { hipLaunchKernelGGL(k_copy_slab, dim3((unsigned)((n4 + len - 1) / len)), dim3(256), 0, 0, a.p[0], a.p[1], n4, len); }); printf("copy slab 64KB/block         %.2f ms %.0f GB/s\n", ms, cb / ms / 1e6); }
    fflush(stdout);
    const double mb = (double)n4 * 16 * 24;
    struct G { int WX, WY, chunk, remap, order; };
    const G gs[] = {{1, 16, 24, 4, 0}, {1, 16, 24, 2, 0}, {1, 16, 24, 8, 0}, {1, 15, 24, 4, 0}, {4, 4, 24, 4, 0}, {1, 16, 24, 1, 0}, {1, 16, 24, 0, 0}, {2, 8, 24, 1, 0}, {4, 4, 24, 1, 0}, {4, 4, 24, 0, 0}, {1, 16, 8, 1, 0}, {4, 4, 8, 1, 0}, {1, 16, 1, 1, 0}, {4, 4, 1, 1, 0}, {4, 4, 1, 0, 0},
                    {1, 16, 24, 0, 1}, {4, 4, 24, 0, 1}, {1, 8, 24, 1, 0}, {1, 4, 24, 1, 0}, {4, 2, 24, 1, 0}, {4, 1, 24, 1, 0}, {1, 16, 64, 1, 0}, {4, 4, 64, 1, 0}};
    for (const G& g : gs) {
        const int nbx = px4 / (64 * g.WX), nby = (ny + g.WY - 1) / g.WY, nbz = (nz + g.chunk - 1) / g.chunk;
        float ms = timeit([&] { hipLaunchKernelGGL((k_march<18, 6>), dim3((unsigned)nbx * nby * nbz), dim3(64 * g.WX * g.WY), 0, 0, a, px4, ny, nz, g.chunk, g.WX, g.WY, nbx, nby, nbz, g.remap, g.order); });
        printf("march 18R+6W  WX=%d WY=%2d chunk=%2d remap=%d order=%d  %.2f ms %.0f GB/s\n", g.WX, g.WY, g.chunk, g.remap, g.order, ms, mb / ms / 1e6); fflush(stdout);
    }
    return 0;
}
