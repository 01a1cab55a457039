// common.hpp -- shared pieces of the gfx950 FDTD engine (host + device).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/fdtd_engine.h"

namespace fdtd {

// ---- error plumbing: no exception ever crosses the C ABI --------------------------------
std::string& last_error_slot();
int fail(int code, const char* fmt, ...);

#define FDTD_HIP(expr)                                                                         \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return ::fdtd::fail(FDTD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// ---- device vocabulary ------------------------------------------------------------------
constexpr int kLanes = 64;        // CDNA wavefront
constexpr int kRowAlign = 256;    // row pitch granule in elements: one wave row of 16-byte lanes (fp32)
constexpr int kMaxLayers = 64;    // CPML layers per face the table storage admits
constexpr int kWavesPerBlock = 4; // 256-thread workgroups: four rows of one x tile

template <typename T> struct VecWidth;                       // elements per 16-byte lane access
template <> struct VecWidth<float> { static constexpr int V = 4; };
template <> struct VecWidth<double> { static constexpr int V = 2; };

template <typename T, int V> struct alignas(16) Pack { T v[V]; };

template <typename T, int V> __device__ __forceinline__ Pack<T, V> ldv(const T* p) {
    return *reinterpret_cast<const Pack<T, V>*>(p);
}
template <typename T, int V> __device__ __forceinline__ void stv(T* p, const Pack<T, V>& x) {
    *reinterpret_cast<Pack<T, V>*>(p) = x;
}
template <typename T, int V> __device__ __forceinline__ Pack<T, V> zero_pack() {
    Pack<T, V> z;
#pragma unroll
    for (int e = 0; e < V; ++e) z.v[e] = (T)0;
    return z;
}

enum Zone : int { Z_NONE = 0, Z_NEG = 1, Z_POS = 2, Z_MID = 3 };

// Zone of index idx along an axis of N cells for a backward-difference (D) term: low CPML
// layers sit at idx = 1+l, high CPML layers at idx = N-np+l, the plain update in between
// (reference sweeps FDTD3D.cpp:540-577).
__host__ __device__ __forceinline__ int zone_e(int idx, int N, int nn, int np, int& l) {
    l = 0;
    if (idx < 1 || idx >= N) return Z_NONE;
    if (idx <= nn) { l = idx - 1; return Z_NEG; }
    if (idx >= N - np) { l = idx - (N - np); return Z_POS; }
    return Z_MID;
}
// Forward-difference (B) term: low CPML at idx = l, high CPML at idx = N-np+l.  `open_skip`
// reproduces the 3-D reference's middle range [nn, min(N-1, N-np)) which leaves the last cell
// untouched when the high face has no CPML (FDTD3D.cpp:822-823); the 2-D code does update it
// (FDTD2D.cpp:408-409).
__host__ __device__ __forceinline__ int zone_h(int idx, int N, int nn, int np, bool open_skip, int& l) {
    l = 0;
    if (idx < 0 || idx >= N) return Z_NONE;
    if (idx < nn) { l = idx; return Z_NEG; }
    if (idx >= N - np) { l = idx - (N - np); return Z_POS; }
    if (open_skip && np == 0 && idx == N - 1) return Z_NONE;
    return Z_MID;
}

// Consecutive workgroup ids are dealt round-robin over the 8 XCDs (each with its own L2).
// This bijection hands every XCD one contiguous range of tiles so that neighbouring tiles,
// which share halo rows and planes, meet in the same L2.
__device__ __forceinline__ unsigned xcd_contiguous(unsigned bid, unsigned nblk) {
    const unsigned q = nblk >> 3, r = nblk & 7u, x = bid & 7u, pos = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + pos;
}

// Counter-based uniform in [0,1): synthetic materials reproducible on host and device.
__host__ __device__ __forceinline__ double hash_unit(uint64_t seed, uint32_t stream, uint64_t index) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (index + 1) + ((uint64_t)stream << 56);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// Per-launch HIP-event timing of the dominant kernels (optional, off by default).
struct KernelTimer {
    bool enabled = false;
    struct Rec { hipEvent_t a, b; int kind; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    hipEvent_t get();
    void begin(int kind, hipStream_t s);
    void end(hipStream_t s);
    int collect(int kind, int* launches, double* total_ms);
    void release();
};

}  // namespace fdtd
