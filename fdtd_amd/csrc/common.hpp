// common.hpp -- shared pieces of the gfx950 FDTD engine (host + device).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <exception>
#include <new>
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

// Makes `want` the calling thread's current device for the lifetime of the object and puts the
// caller's device back afterwards, so engines on several devices can share one host thread.
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    hipError_t status = hipSuccess;
    explicit DeviceScope(int want) {
        status = hipGetDevice(&prev);
        if (status == hipSuccess && want >= 0 && want != prev) {
            status = hipSetDevice(want);
            switched = status == hipSuccess;
        }
    }
    ~DeviceScope() { if (switched) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};

// Body of an entry point that takes a handle: null check, the engine's device made current, and
// any C++ exception of the host side (a failed staging allocation) turned into a status code.
template <typename H, typename F>
int entry(H h, F&& body) {
    if (!h) return fail(FDTD_ERR_ARG, "null handle");
    try {
        DeviceScope scope(h->device);
        if (scope.status != hipSuccess) return fail(FDTD_ERR_HIP, "cannot select device %d: %s", h->device, hipGetErrorString(scope.status));
        return body();
    } catch (const std::bad_alloc&) {
        return fail(FDTD_ERR_HOST, "out of host memory");
    } catch (const std::exception& e) {
        return fail(FDTD_ERR_HOST, "%s", e.what());
    }
}

// ---- device vocabulary ------------------------------------------------------------------
#define FDTD_AS(n) __attribute__((address_space(n)))   // 1: global, 4: constant (read through the scalar cache)
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

// The value the neighbouring lane holds: lane i receives lane i-1's (`lane_below`) or lane i+1's (`lane_above`); the
// first (last) lane keeps its own.  One data-parallel-primitive move across the whole wave (wave_shr:1 / wave_shl:1)
// instead of a permute through the LDS unit, which the row exchanges of the kernel keep busy enough.
template <int CTRL>
__device__ __forceinline__ int wave_shift_bits(int x) { return __builtin_amdgcn_update_dpp(x, x, CTRL, 0xf, 0xf, false); }
template <int CTRL, typename T>
__device__ __forceinline__ T wave_shift(T x) {
    if constexpr (sizeof(T) == 4) {
        return __builtin_bit_cast(T, wave_shift_bits<CTRL>(__builtin_bit_cast(int, x)));
    } else {
        static_assert(sizeof(T) == 8, "32- or 64-bit values");
        struct Two { int lo, hi; } t = __builtin_bit_cast(Two, x);
        t.lo = wave_shift_bits<CTRL>(t.lo);
        t.hi = wave_shift_bits<CTRL>(t.hi);
        return __builtin_bit_cast(T, t);
    }
}
template <typename T> __device__ __forceinline__ T lane_below(T x) { return wave_shift<0x138>(x); }
template <typename T> __device__ __forceinline__ T lane_above(T x) { return wave_shift<0x130>(x); }

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

// In between: runs of G consecutive tiles stay on one XCD, successive runs go round the XCDs, so that
// all XCDs stream from the same neighbourhood of memory while the tiles of a run still meet in one L2.
__device__ __forceinline__ unsigned xcd_grouped(unsigned bid, unsigned nblk, unsigned G) {
    const unsigned span = 8u * G, bulk = nblk / span * span;
    if (bid >= bulk) return bid;                       // the tail keeps the order it is dealt in
    const unsigned x = bid & 7u, pos = bid >> 3;
    return ((pos / G) * 8u + x) * G + pos % G;
}

// Counter-based uniform in [0,1): synthetic materials reproducible on host and device.
__host__ __device__ __forceinline__ double hash_unit(uint64_t seed, uint32_t stream, uint64_t index) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (index + 1) + ((uint64_t)stream << 56);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

// ---- probe time series: a handful of cells sampled after every step without leaving the device ----
struct ProbeSpec { int num, den; long long off; double sdiv; };   // value = arr[num][off], divided by arr[den][off] (den >= 0) or by sdiv (den == -2)
template <typename T, int N> struct PtrTable { const T* p[N]; };
template <typename T, int N>
__global__ void k_probe(PtrTable<T, N> t, const ProbeSpec* __restrict__ ps, int n_probes, T* __restrict__ row) {
    const int q = threadIdx.x;
    if (q >= n_probes) return;
    const ProbeSpec s = ps[q];
    T v = t.p[s.num][s.off];
    if (s.den >= 0) v = v / t.p[s.den][s.off];
    else if (s.den == -2) v = v / (T)s.sdiv;
    row[q] = v;
}
constexpr int kMaxProbes = 64;

// Host-side bookkeeping of the sample buffer ([sample][probe], grown by doubling).
template <typename T>
struct ProbeLog {
    std::vector<ProbeSpec> specs;
    ProbeSpec* specs_d = nullptr;
    T* buf = nullptr;
    int rows = 0, cap = 0;
    bool dirty = true;
    int add(const ProbeSpec& s);
    int next_row(T** row, int want_cap);        // makes room, uploads specs if needed, returns the row to fill
    int read(double* out, size_t cap_values, int* n_samples, int* n_probes);
    void reset(bool drop_probes) { rows = 0; if (drop_probes) { specs.clear(); dirty = true; } }
    void release();
};
template <typename T> int ProbeLog<T>::add(const ProbeSpec& s) {
    if ((int)specs.size() >= kMaxProbes) return fail(FDTD_ERR_ARG, "at most %d probes", kMaxProbes);
    if (rows > 0) return fail(FDTD_ERR_STATE, "probes must be added before the first sample is recorded (or after a reset)");
    specs.push_back(s);
    dirty = true;
    return FDTD_OK;
}
template <typename T> int ProbeLog<T>::next_row(T** row, int want_cap) {
    if (dirty) {
        if (!specs_d) FDTD_HIP(hipMalloc((void**)&specs_d, kMaxProbes * sizeof(ProbeSpec)));
        FDTD_HIP(hipMemcpy(specs_d, specs.data(), specs.size() * sizeof(ProbeSpec), hipMemcpyHostToDevice));
        dirty = false;
    }
    if (rows >= cap) {
        const int ncap = std::max(std::max(want_cap, 64), cap * 2);
        T* nb = nullptr;
        FDTD_HIP(hipMalloc((void**)&nb, (size_t)ncap * kMaxProbes * sizeof(T)));
        if (buf) {
            FDTD_HIP(hipDeviceSynchronize());
            FDTD_HIP(hipMemcpy(nb, buf, (size_t)rows * kMaxProbes * sizeof(T), hipMemcpyDeviceToDevice));
            (void)hipFree(buf);
        }
        buf = nb; cap = ncap;
    }
    *row = buf + (size_t)rows * kMaxProbes;
    ++rows;
    return FDTD_OK;
}
template <typename T> int ProbeLog<T>::read(double* out, size_t cap_values, int* n_samples, int* n_probes) {
    const int np = (int)specs.size();
    if (n_samples) *n_samples = rows;
    if (n_probes) *n_probes = np;
    if (!out) return FDTD_OK;
    if (cap_values < (size_t)rows * np) return fail(FDTD_ERR_ARG, "probe_read: room for %zu values, %zu recorded", cap_values, (size_t)rows * np);
    if (rows == 0 || np == 0) return FDTD_OK;
    std::vector<T> st((size_t)rows * kMaxProbes);
    FDTD_HIP(hipDeviceSynchronize());
    FDTD_HIP(hipMemcpy(st.data(), buf, st.size() * sizeof(T), hipMemcpyDeviceToHost));
    for (int r = 0; r < rows; ++r) for (int q = 0; q < np; ++q) out[(size_t)r * np + q] = (double)st[(size_t)r * kMaxProbes + q];
    return FDTD_OK;
}
template <typename T> void ProbeLog<T>::release() {
    if (buf) (void)hipFree(buf);
    if (specs_d) (void)hipFree(specs_d);
    buf = nullptr; specs_d = nullptr;
}

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// Per-launch HIP-event timing of the dominant kernels (optional, off by default).
struct KernelTimer {
    bool enabled = false;
    struct Rec { hipEvent_t a, b; int kind, part; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    hipEvent_t get();
    void begin(int kind, hipStream_t s, int part = 0);
    void end(hipStream_t s);
    int collect(int kind, int* launches, double* total_ms);
    // every launch recorded since the last collect(), in launch order: kind, part, start relative to the
    // first one's start and duration (ms); returns the number of launches (which may exceed `cap`)
    int timeline(int cap, int* kinds, int* parts, double* start_ms, double* dur_ms, int* n_out);
    void release();
};

}  // namespace fdtd
