// HBM streaming ceilings for multi-array access patterns (diagnostic, not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Ptrs { float4* p[24]; };

// grid-stride: each thread handles float4 at index; R arrays read, W arrays written
template <int R, int W>
__global__ void __launch_bounds__(256) k_stream(Ptrs a, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < R; ++r) { float4 v = a.p[r][i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
#pragma unroll
        for (int w = 0; w < W; ++w) { float4 o = acc; o.x += w; a.p[R + w][i] = o; }
    }
}
// plane-march: block (64 lanes x 4 rows) marches `chunk` planes; same geometry as the FDTD kernels
template <int R, int W, int NT>
__global__ void __launch_bounds__(256) k_march(Ptrs a, int px4, int ny, int nz, int chunk, int nbx, int nby) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned bid = blockIdx.x;
    const int bx = bid % nbx; bid /= nbx; const int by = bid % nby; const int bz = bid / nby;
    const int j = by * 4 + w; if (j >= ny) return;
    const int k0 = bz * chunk, k1 = min(k0 + chunk, nz);
    size_t idx = ((size_t)k0 * ny + j) * px4 + bx * 64 + lane;
    const size_t sz = (size_t)ny * px4;
    for (int k = k0; k < k1; ++k, idx += sz) {
        float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < R; ++r) { float4 v; if (NT & 1) { v4f t = __builtin_nontemporal_load((const v4f*)&a.p[r][idx]); v = make_float4(t.x, t.y, t.z, t.w); } else v = a.p[r][idx]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
#pragma unroll
        for (int wv = 0; wv < W; ++wv) { float4 o = acc; o.x += wv; if (NT & 2) { v4f t = {o.x, o.y, o.z, o.w}; __builtin_nontemporal_store(t, (v4f*)&a.p[R + wv][idx]); } else a.p[R + wv][idx] = o; }
    }
}
template <int R, int W, int NT = 0> int run(const char* name, Ptrs a, size_t n4, int px4, int ny, int nz, bool march, int chunk) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        if (march) { int nbx = px4 / 64, nby = (ny + 3) / 4, nbz = (nz + chunk - 1) / chunk; hipLaunchKernelGGL((k_march<R, W, NT>), dim3(nbx * nby * nbz), dim3(256), 0, 0, a, px4, ny, nz, chunk, nbx, nby); }
        else hipLaunchKernelGGL((k_stream<R, W>), dim3(256 * 16), dim3(256), 0, 0, a, n4);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double bytes = (double)n4 * 16 * (R + W);
    printf("%-28s R=%2d W=%d  %.2f ms  %.0f GB/s\n", name, R, W, best, bytes / best / 1e6); fflush(stdout);
    return 0;
}
int main() {
    const int px = 1024, ny = 1025, nz = 1026; const int px4 = px / 4;
    const size_t n4 = (size_t)px4 * ny * nz;
    Ptrs a;
    for (int i = 0; i < 24; ++i) { CK(hipMalloc((void**)&a.p[i], n4 * 16 + 4096)); CK(hipMemset(a.p[i], 0, n4 * 16)); }
    CK(hipDeviceSynchronize());
    run<1, 1>("stream copy", a, n4, px4, ny, nz, false, 0);
    run<2, 1>("stream", a, n4, px4, ny, nz, false, 0);
    run<9, 6>("stream (E-half full mix)", a, n4, px4, ny, nz, false, 0);
    run<6, 3>("stream (E-half lean mix)", a, n4, px4, ny, nz, false, 0);
    run<12, 3>("stream (H-half lean+mu mix)", a, n4, px4, ny, nz, false, 0);
    run<15, 0>("stream read only", a, n4, px4, ny, nz, false, 0);
    run<1, 6>("stream write mostly", a, n4, px4, ny, nz, false, 0);
    run<9, 6>("march chunk16 (E full)", a, n4, px4, ny, nz, true, 16);
    run<6, 3>("march chunk16 (E lean)", a, n4, px4, ny, nz, true, 16);
    run<9, 3>("march chunk16 (H lean)", a, n4, px4, ny, nz, true, 16);
    run<6, 3, 1>("march16 lean NT loads", a, n4, px4, ny, nz, true, 16);
    run<6, 3, 2>("march16 lean NT stores", a, n4, px4, ny, nz, true, 16);
    run<6, 3, 3>("march16 lean NT both", a, n4, px4, ny, nz, true, 16);
    run<9, 6, 3>("march16 full NT both", a, n4, px4, ny, nz, true, 16);
    run<6, 3>("march chunk64 (E lean)", a, n4, px4, ny, nz, true, 64);
    run<6, 3>("march chunk4 (E lean)", a, n4, px4, ny, nz, true, 4);
    run<6, 3>("march chunk1 (E lean)", a, n4, px4, ny, nz, true, 1);
    return 0;
}
