// Calibration of the FETCH_SIZE counter for the access shapes of the two-step 3-D kernel (8 bytes per lane, 512-byte
// windows that start every 496 bytes) against shapes whose byte count is known.  Every kernel reads a 2 GiB buffer
// (far beyond the 256 MiB Infinity Cache) once; the name of the kernel says which bytes it touches.
//
//     hipcc -O3 --offload-arch=gfx950 tools/calib_fetch.hip -o tools/calib_fetch
//     rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- tools/calib_fetch
//
// Prints, per kernel, the bytes the lanes asked for, the distinct 128-byte lines, 64-byte and 32-byte blocks those
// bytes lie in, and the time of a launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr size_t kBytes = size_t(2) << 30;
constexpr int kWavesPerBlock = 4;

__device__ __forceinline__ void keep(float s, float* sink) {
    if (s == 123.456f) sink[0] = s;       // (never true on a zeroed buffer; keeps the loads alive)
}

// 16 bytes per lane, consecutive lanes consecutive: the shape the counter is documented for
__global__ void calib_stream16(const float4* p, size_t n, float* sink) {
    float s = 0;
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) { float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
    keep(s, sink);
}
// 8 bytes per lane, consecutive lanes consecutive (512 bytes per wave instruction, aligned)
__global__ void calib_stream8(const float2* p, size_t n, float* sink) {
    float s = 0;
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) { float2 v = p[i]; s += v.x + v.y; }
    keep(s, sink);
}
// rows of 4096 bytes; a wave reads a 512-byte window (8 bytes per lane) that starts 496*t bytes into its row, t = 0..8:
// the windows of neighbouring tiles share 16 bytes and start on no line boundary.  Waves of different t are in
// different workgroups, as the x tiles of the two-step kernel are.
// SAME_XCD: the nine tiles of a row group are nine consecutive workgroups of ONE XCD (workgroup b runs on XCD b % 8).
template <bool SAME_XCD>
__global__ void calib_window496(const float* p, size_t rows, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = SAME_XCD ? (blockIdx.x / 8 / 9 * 8 + blockIdx.x % 8) * 9 + blockIdx.x / 8 % 9 : blockIdx.x;
    const int t = b % 9;
    float s = 0;
    for (size_t r = (b / 9) * size_t(kWavesPerBlock) + wave; r < rows; r += size_t(gridDim.x / 9) * kWavesPerBlock) {
        const float* row = p + r * 1024;
        int c = 124 * t + 2 * lane;
        if (c + 1 < 1024) { float2 v = *reinterpret_cast<const float2*>(row + c); s += v.x + v.y; }
    }
    keep(s, sink);
}
// the same windows cut at 512 bytes (aligned, no overlap): what the tiles would cost without halo columns
__global__ void calib_window512(const float* p, size_t rows, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = blockIdx.x % 8;
    float s = 0;
    for (size_t r = (blockIdx.x / 8) * size_t(kWavesPerBlock) + wave; r < rows; r += size_t(gridDim.x / 8) * kWavesPerBlock) {
        const float* row = p + r * 1024;
        float2 v = *reinterpret_cast<const float2*>(row + 128 * t + 2 * lane);
        s += v.x + v.y;
    }
    keep(s, sink);
}
// 8 bytes out of every STRIDE bytes
template <int STRIDE>
__global__ void calib_sparse8(const char* p, size_t n, float* sink) {
    float s = 0;
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        float2 v = *reinterpret_cast<const float2*>(p + i * STRIDE);
        s += v.x + v.y;
    }
    keep(s, sink);
}

// R arrays read and W arrays written per cell, 8 bytes per lane, non-temporal stores: the mix of the two-step kernel
// (12 arrays in, 6 out) without any arithmetic to speak of
struct MixPtrs { const float2* in[12]; float2* out[6]; };
template <int R, int W>
__global__ void calib_mix(MixPtrs p, size_t n) {
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        float2 s = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < R; ++r) { const float2 v = p.in[r][i]; s.x += v.x; s.y += v.y; }
#pragma unroll
        for (int w = 0; w < W; ++w) {
            typedef float vec2 __attribute__((ext_vector_type(2)));
            vec2 o = {s.x + float(w), s.y};
            __builtin_nontemporal_store(o, reinterpret_cast<vec2*>(&p.out[w][i]));
        }
    }
}

template <typename F>
static void timed(const char* name, double asked, double l128, double b64, double b32, F launch) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < 3; ++i) launch();
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    std::printf("%-24s asked %8.1f MB  lines128 %8.1f MB  blocks64 %8.1f MB  blocks32 %8.1f MB  %7.3f ms  (%6.0f GB/s of asked bytes)\n", name, asked / 1e6,
                l128 / 1e6, b64 / 1e6, b32 / 1e6, ms, asked / ms / 1e6);
}

int main() {
    char* buf;
    float* sink;
    CHECK(hipMalloc(&buf, kBytes));
    CHECK(hipMalloc(&sink, 256));
    CHECK(hipMemset(buf, 0, kBytes));
    const int threads = 64 * kWavesPerBlock;
    const double B = double(kBytes);
    timed("calib_stream16", B, B, B, B, [&] { calib_stream16<<<256 * 16, threads>>>(reinterpret_cast<const float4*>(buf), kBytes / 16, sink); });
    timed("calib_stream8", B, B, B, B, [&] { calib_stream8<<<256 * 16, threads>>>(reinterpret_cast<const float2*>(buf), kBytes / 8, sink); });
    const size_t rows = kBytes / 4096;
    // tiles 0..7 read 512 bytes, tile 8 the last 128 bytes of the row: 9 windows cover columns 0..1023 with 8 x 16 bytes read twice
    const double asked496 = double(rows) * (8 * 512 + (1024 - 992) * 4);
    // lines touched per row, tiles in different workgroups: tile t touches lines floor(496t/128) .. floor((496t+511)/128)
    double lines = 0, b64 = 0, b32 = 0;
    for (int t = 0; t < 9; ++t) {
        int lo = 496 * t, hi = (t < 8 ? lo + 511 : 4095);
        lines += hi / 128 - lo / 128 + 1;
        b64 += hi / 64 - lo / 64 + 1;
        b32 += hi / 32 - lo / 32 + 1;
    }
    timed("calib_window496<false>", asked496, rows * lines * 128, rows * b64 * 64, rows * b32 * 32,
          [&] { calib_window496<false><<<9 * 256 * 4, threads>>>(reinterpret_cast<const float*>(buf), rows, sink); });
    timed("calib_window496<true>", asked496, rows * lines * 128, rows * b64 * 64, rows * b32 * 32,
          [&] { calib_window496<true><<<9 * 256 * 4, threads>>>(reinterpret_cast<const float*>(buf), rows, sink); });
    timed("calib_window512", B, B, B, B, [&] { calib_window512<<<8 * 256 * 4, threads>>>(reinterpret_cast<const float*>(buf), rows, sink); });
    timed("calib_sparse8<128>", B / 16, B, B / 2, B / 4, [&] { calib_sparse8<128><<<256 * 16, threads>>>(buf, kBytes / 128, sink); });
    timed("calib_sparse8<64>", B / 8, B, B, B / 2, [&] { calib_sparse8<64><<<256 * 16, threads>>>(buf, kBytes / 64, sink); });
    timed("calib_sparse8<32>", B / 4, B, B, B, [&] { calib_sparse8<32><<<256 * 16, threads>>>(buf, kBytes / 32, sink); });
    {
        // 18 arrays of 256 MiB, each at the start of a 1 GiB allocation of its own
        MixPtrs mp;
        void* blocks[18];
        const size_t each = size_t(256) << 20;
        for (int q = 0; q < 18; ++q) { CHECK(hipMalloc(&blocks[q], size_t(1) << 30)); CHECK(hipMemset(blocks[q], 0, each)); }
        for (int q = 0; q < 12; ++q) mp.in[q] = reinterpret_cast<const float2*>(blocks[q]);
        for (int q = 0; q < 6; ++q) mp.out[q] = reinterpret_cast<float2*>(blocks[12 + q]);
        const double E = double(each);
        timed("calib_mix<12,6>", 18 * E, 18 * E, 18 * E, 18 * E, [&] { calib_mix<12, 6><<<256 * 16, threads>>>(mp, each / 8); });
        timed("calib_mix<6,6>", 12 * E, 12 * E, 12 * E, 12 * E, [&] { calib_mix<6, 6><<<256 * 16, threads>>>(mp, each / 8); });
        timed("calib_mix<12,0>", 12 * E, 12 * E, 12 * E, 12 * E, [&] { calib_mix<12, 0><<<256 * 16, threads>>>(mp, each / 8); });
        CHECK(hipDeviceSynchronize());
        for (int q = 0; q < 18; ++q) CHECK(hipFree(blocks[q]));
    }
    CHECK(hipDeviceSynchronize());
    CHECK(hipFree(buf));
    CHECK(hipFree(sink));
    return 0;
}
