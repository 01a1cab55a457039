// Can the arrays of one compact virtual range be SPREAD over the device's physical memory without holding the memory
// in between?  hipcc --offload-arch=gfx950 -O3 -o stream_spread tools/stream_spread.hip && ./stream_spread [planes]
// 24 arrays of 1024 x 1025 x planes fp32, adjacent in one reserved virtual range, timed with the 18-read + 6-write
// plane march of tools/stream_lottery.hip for three ways of backing the range:
//   malloc      one hipMalloc (arrays adjacent in physical memory as far as the driver keeps them so)
//   vmm 0       128 MiB physical chunks created one after the other and mapped in order (same, through the VMM API)
//   vmm G       `G` GiB worth of chunks created between two arrays and released again right away: if the driver hands
//               physical memory out in order, the arrays end up G GiB apart physically while the process only holds
//               the memory of the arrays
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Lay { float4* p[24]; };

__device__ __forceinline__ unsigned xcd_grouped(unsigned bid, unsigned nblk, unsigned G) {
    const unsigned span = 8u * G, bulk = nblk / span * span;
    if (bid >= bulk) return bid;
    const unsigned x = bid & 7u, pos = bid >> 3;
    return ((pos / G) * 8u + x) * G + pos % G;
}
// BAR: a workgroup barrier per plane (as the fused kernel has); DIV: IEEE divisions per loaded value pair (the fused
// kernel divides 8 packs per plane: H = B/mu five times, E = D/eps three times)
template <int R, int W, int BAR, int DIV>
__global__ void __launch_bounds__(1024) k_march(Lay a, int px4, int ny, int nz, int chunk, int nbx, int nby) {
    const int lane = threadIdx.x & 63, wy = threadIdx.x >> 6;
    unsigned bid = xcd_grouped(blockIdx.x, gridDim.x, 4);
    const int bx = bid % nbx; bid /= nbx;
    const int by = bid % nby, bz = bid / nby;
    const int j = min(by * 16 + wy, ny - 1);
    const int k0 = bz * chunk, k1 = min(k0 + chunk, nz);
    size_t idx = ((size_t)k0 * ny + j) * px4 + bx * 64 + lane;
    for (int k = k0; k < k1; ++k, idx += (size_t)ny * px4) {
        float4 acc = make_float4(0, 0, 0, 0);
        float4 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = a.p[r][idx];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (r < DIV) { const float4 m = v[(r + 1) % R]; v[r].x = v[r].x / (m.x + 2.f); v[r].y = v[r].y / (m.y + 2.f); v[r].z = v[r].z / (m.z + 2.f); v[r].w = v[r].w / (m.w + 2.f); }
            acc.x += v[r].x; acc.y += v[r].y; acc.z += v[r].z; acc.w += v[r].w;
        }
        if (BAR) __syncthreads();
#pragma unroll
        for (int w = 0; w < W; ++w) { v4f t = {acc.x + w, acc.y, acc.z, acc.w}; __builtin_nontemporal_store(t, (v4f*)&a.p[R + w][idx]); }
    }
}

int main(int argc, char** argv) {
    const int px = 1024, ny = 1025, nz = argc > 1 ? atoi(argv[1]) : 342, px4 = px / 4;
    const size_t chunk = (size_t)128 << 20;
    const size_t arr = ((size_t)px * ny * nz * 4 + 4095) / 4096 * 4096;
    const size_t total = (24 * arr + chunk - 1) / chunk * chunk;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    printf("24 arrays of %.2f GiB; march 18R+6W, ms (best of 3) and GB/s\n", arr / 1073741824.0);
    for (int mode : {-1, 6, 8}) {
        char* base = nullptr;
        std::vector<hipMemGenericAllocationHandle_t> kept;
        hipEventRecord(e0);
        if (mode < 0) { CK(hipMalloc((void**)&base, total)); }
        else {
            CK(hipMemAddressReserve((void**)&base, total, chunk, nullptr, 0));
            const size_t gap_chunks = ((size_t)mode << 30) / chunk;
            std::vector<hipMemGenericAllocationHandle_t> dropped;
            int cur = 0;
            for (size_t off = 0; off < total; off += chunk) {
                const int owner = (int)std::min<size_t>(off / arr, 23);       // the array this chunk starts in
                if (owner != cur) {                                            // memory to leave out in front of the next array
                    for (size_t g = 0; g < gap_chunks; ++g) { hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, chunk, &prop, 0)); dropped.push_back(h); }
                    cur = owner;
                }
                hipMemGenericAllocationHandle_t h;
                CK(hipMemCreate(&h, chunk, &prop, 0));
                CK(hipMemMap(base + off, chunk, 0, h, 0));
                kept.push_back(h);
            }
            for (auto h : dropped) CK(hipMemRelease(h));
            CK(hipMemSetAccess(base, total, &acc, 1));
        }
        CK(hipMemset(base, 0, total));
        hipEventRecord(e1); hipEventSynchronize(e1);
        float setup_ms; hipEventElapsedTime(&setup_ms, e0, e1);
        size_t free_b = 0, tot_b = 0; CK(hipMemGetInfo(&free_b, &tot_b));
        Lay a; for (int q = 0; q < 24; ++q) a.p[q] = (float4*)(base + (size_t)q * arr);
        const int nbx = px4 / 64, nby = (ny + 15) / 16, nbz = (nz + 23) / 24;
        auto timeit = [&](auto kern) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(kern, dim3((unsigned)nbx * nby * nbz), dim3(1024), 0, 0, a, px4, ny, nz, 24, nbx, nby);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            return best;
        };
        const float t0 = timeit(k_march<18, 6, 0, 0>), t1 = timeit(k_march<18, 6, 1, 0>), t2 = timeit(k_march<18, 6, 0, 8>), t3 = timeit(k_march<18, 6, 1, 8>);
        CK(hipGetLastError());
        if (mode < 0) printf("malloc      ");
        else printf("vmm %dG gaps ", mode);
        const double gb = (double)arr * 24 / 1e6;
        printf("plain %6.2f ms %5.0f GB/s | barrier %5.0f | 8 divisions %5.0f | both %5.0f GB/s   (set-up %.0f ms, in use %.1f GiB)\n", t0, gb / t0, gb / t1, gb / t2, gb / t3, setup_ms, (tot_b - free_b) / 1073741824.0);
        fflush(stdout);
        if (mode < 0) CK(hipFree(base));
        else { CK(hipMemUnmap(base, total)); for (auto h : kept) CK(hipMemRelease(h)); CK(hipMemAddressFree(base, total)); }
    }
    return 0;
}
