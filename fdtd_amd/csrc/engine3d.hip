// engine3d.hip -- host side of the 3-D engine: device state, array exchange, launches, C ABI.
#include "kernels3d_fused.hpp"
#include "kernels3d_pair.hpp"
#include "../../include/fdtd_setup.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>

namespace fdtd {

// Order-independent 64-bit checksum term of one value at one canonical index.
__host__ __device__ __forceinline__ uint64_t mix_term(uint64_t bits, uint64_t index) {
    uint64_t z = bits + 0x9E3779B97F4A7C15ull * (index + 0x632BE59BD9B4E019ull);
    z = (z ^ (z >> 32)) * 0xD6E8FEB86659FD93ull;
    z = (z ^ (z >> 32)) * 0xD6E8FEB86659FD93ull;
    return z ^ (z >> 32);
}

// Geometry of one stored component for the utility kernels.
struct CompGeom {
    int n[3];          // canonical extents
    int Nx, px, ny1;   // stored columns, pitch, rows
    int kl_lo, kl_hi;  // owned planes [kl_lo, kl_hi)
    int zoff;
};

template <typename T>
__global__ void k_fill_hash(T* arr, CompGeom g, uint64_t seed, uint32_t stream, double lo, double hi, int kl_all_lo, int kl_all_hi) {
    // one thread per stored element of the planes [kl_all_lo, kl_all_hi) (owned + ghosts)
    const size_t per_plane = (size_t)g.ny1 * g.px;
    const size_t total = per_plane * (size_t)(kl_all_hi - kl_all_lo);
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int kl = kl_all_lo + (int)(t / per_plane);
        const size_t r = t % per_plane;
        const int j = (int)(r / g.px), i = (int)(r % g.px);
        const int k = kl + g.zoff;
        if (i >= g.Nx || i >= g.n[0] || j >= g.n[1] || k < 0 || k >= g.n[2]) continue;
        const uint64_t canon = ((uint64_t)i * g.n[1] + j) * g.n[2] + k;
        arr[(size_t)kl * per_plane + r] = (T)(lo + (hi - lo) * hash_unit(seed, stream, canon));
    }
}

// FDTD3D::setupMaterials (FDTD3D.cpp:106-200) evaluated per stored element in fp64, operation for
// operation as the host generator fdtd3d_default_materials does it: component c sits half a cell along
// its own axis, the other two axes are sampled at the node and half a cell either side (3 x 3 points);
// the fraction of points inside the sphere blends eps_air into eps_glass.
template <typename T>
__global__ void k_fill_sphere(T* arr, CompGeom g, int c, int N0, int N1, int N2, double h0, double h1, double h2,
                              double eps_glass, double eps_air, double radius, int kl_all_lo, int kl_all_hi) {
    const size_t per_plane = (size_t)g.ny1 * g.px;
    const size_t total = per_plane * (size_t)(kl_all_hi - kl_all_lo);
    const int N[3] = {N0, N1, N2};
    const double h[3] = {h0, h1, h2};
    const int u = (c + 1) % 3 < (c + 2) % 3 ? (c + 1) % 3 : (c + 2) % 3;
    const int v = 3 - c - u;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int kl = kl_all_lo + (int)(t / per_plane);
        const size_t r = t % per_plane;
        const int p[3] = {(int)(r % g.px), (int)(r / g.px), kl + g.zoff};
        if (p[0] >= g.Nx || p[0] >= g.n[0] || p[1] >= g.n[1] || p[2] < 0 || p[2] >= g.n[2]) continue;
        double r0[3];
        for (int a = 0; a < 3; ++a) r0[a] = a == c ? ((p[a] + 0.5) - N[a] / 2) * h[a] : (p[a] - N[a] / 2) * h[a];
        int hits = 0;
        for (int su = -1; su <= 1; ++su) for (int sv = -1; sv <= 1; ++sv) {
            double q[3] = {r0[0], r0[1], r0[2]};
            q[u] = r0[u] + su * (h[u] * 0.5);
            q[v] = r0[v] + sv * (h[v] * 0.5);
            if (sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]) <= radius) ++hits;
        }
        const double fill = (double)hits / 9.0;
        arr[(size_t)kl * per_plane + r] = (T)(eps_glass + (eps_air - eps_glass) * fill);
    }
}

// Array exchange for field-shaped arrays: the host order is [i][j][k] with k unit-stride, the device
// order [plane][row][column] with i unit-stride.  One workgroup transposes a 32 (k) x 32 (i) tile of
// row j through LDS so that both sides are accessed in their unit-stride direction.  `stage` holds
// the host's values for columns [i0, i0+ci), all j, all k.  Wall columns (i >= Nx) and planes the
// slab does not store are skipped.
struct ExchangeGeom {
    int n1, n2;        // canonical extents along j and k
    int i0, ci;        // chunk of columns in the staging buffer
    int Nx, px, ny1, nzp, zoff;
};
template <typename T>
__global__ void __launch_bounds__(256) k_scatter_field(const double* __restrict__ stage, T* __restrict__ dev, ExchangeGeom g) {
    __shared__ T tile[32][33];
    const int j = blockIdx.y, k0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int il = ty; il < 32; il += 8) {
        const int k = k0 + tx;
        if (il < g.ci && k < g.n2) tile[il][tx] = (T)stage[((size_t)il * g.n1 + j) * g.n2 + k];
    }
    __syncthreads();
    for (int kk = ty; kk < 32; kk += 8) {
        const int k = k0 + kk, kl = k - g.zoff, i = g.i0 + tx;
        if (tx < g.ci && k < g.n2 && kl >= 0 && kl < g.nzp && i < g.Nx) dev[((size_t)kl * g.ny1 + j) * g.px + i] = tile[tx][kk];
    }
}
// The other direction; `den` (or the scalar `sdiv` when den is null and sdiv != 0) divides on the way
// out, which is how E = D/eps and H = B/mu are reported from lean storage.  Positions the slab does
// not store come back as 0 and are not copied to the caller by the host side.
template <typename T>
__global__ void __launch_bounds__(256) k_gather_field(double* __restrict__ stage, const T* __restrict__ dev, const T* __restrict__ den, T sdiv, ExchangeGeom g) {
    __shared__ T tile[32][33];
    const int j = blockIdx.y, k0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int kk = ty; kk < 32; kk += 8) {
        const int k = k0 + kk, kl = k - g.zoff, i = g.i0 + tx;
        T v = (T)0;
        if (tx < g.ci && k < g.n2 && kl >= 0 && kl < g.nzp && i < g.Nx) {
            const size_t off = ((size_t)kl * g.ny1 + j) * g.px + i;
            v = dev[off];
            if (den) v = v / den[off];
            else if (sdiv != (T)0) v = v / sdiv;
        }
        tile[tx][kk] = v;
    }
    __syncthreads();
    for (int il = ty; il < 32; il += 8) {
        const int k = k0 + tx;
        if (il < g.ci && k < g.n2) stage[((size_t)il * g.n1 + j) * g.n2 + k] = (double)tile[il][tx];
    }
}

template <typename T>
__global__ void k_fill_const(T* arr, size_t n, T v) {
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) arr[t] = v;
}

template <typename T> struct BitsOf;
template <> struct BitsOf<float> { __host__ __device__ static uint64_t get(float v) { if (v == 0.0f) v = 0.0f; union { float f; uint32_t u; } x; x.f = v; return x.u; } };
template <> struct BitsOf<double> { __host__ __device__ static uint64_t get(double v) { if (v == 0.0) v = 0.0; union { double f; uint64_t u; } x; x.f = v; return x.u; } };

// out[0] += sum of squares (double), out[1] += checksum, over owned cells.
template <typename T>
__global__ void k_reduce(const T* arr, CompGeom g, double* sumsq_partial, unsigned long long* checksum) {
    const size_t per_plane = (size_t)g.ny1 * g.px;
    const size_t total = per_plane * (size_t)(g.kl_hi - g.kl_lo);
    double s = 0;
    unsigned long long c = 0;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int kl = g.kl_lo + (int)(t / per_plane);
        const size_t r = t % per_plane;
        const int j = (int)(r / g.px), i = (int)(r % g.px);
        const int k = kl + g.zoff;
        if (i >= g.Nx || i >= g.n[0] || j >= g.n[1] || k < 0 || k >= g.n[2]) continue;
        const T v = arr[(size_t)kl * per_plane + r];
        s += (double)v * (double)v;
        if (v != (T)0) c += mix_term(BitsOf<T>::get(v), ((uint64_t)i * g.n[1] + j) * g.n[2] + k);
    }
    __shared__ double sh[256];
    __shared__ unsigned long long shc[256];
    sh[threadIdx.x] = s; shc[threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { sh[threadIdx.x] += sh[threadIdx.x + o]; shc[threadIdx.x] += shc[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { sumsq_partial[blockIdx.x] = sh[0]; atomicAdd(checksum, shc[0]); }
}

// ---- fdtd3d_reduce_static: the reduction above without any allocation ------------------------------
// Its partial sums live in this file-scope array instead of memory obtained from hipMalloc, so the
// call can be made where allocating is not an option.  Every GPU the library runs on holds its own copy
// of the array (at its own address) until the process ends; all calls on one GPU share that copy, so
// calls that overlap on one GPU race.
constexpr int kStaticMaxIds = 16;                     // arrays reduced by one call
constexpr int kReduceBlocks = 1024;                   // workgroups of k_reduce = partial sums per array
constexpr int kStaticSlot = kReduceBlocks + 1;        // per array: the partial sums, then the checksum word
constexpr size_t kStaticDoubles = (size_t)kStaticMaxIds * kStaticSlot + 2 * kStaticMaxIds;   // + (sum, checksum) per array
static __device__ __align__(256) double scratch[kStaticDoubles];       // 131,456 bytes

// One thread per array: adds the partial sums in index order (the order the host loop of the allocating
// entry point uses, so both give the same bits) and packs (sum, checksum bits) for one small download.
__global__ void k_reduce_finish(const double* slots, double* results, int n_ids) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_ids) return;
    const double* sl = slots + (size_t)q * kStaticSlot;
    double acc = 0;
    for (int b = 0; b < kReduceBlocks; ++b) acc += sl[b];
    results[2 * q] = acc;
    results[2 * q + 1] = sl[kReduceBlocks];
}

// Cell-centre observables (FDTD3D::run / dumpFrame, FDTD3D.cpp:1023-1097): every cell averages the
// four Ex, Ey, Ez samples on its edges.  MODE 0: acc (+)= (1/16)(sx^2 + sy^2 + sz^2) * weight, the
// running RMS window, kept [plane][row][column] like a field array (the two-step kernel advances the same
// array by itself when asked to); MODE 1: out0 = (1/4) sx (centred Ex), out1 = ax^2 + ay^2 + az^2 (|E|^2),
// x fastest, then y, then z -- the order the reference hands to its VTI writer.
template <typename T, bool LEAN, int MODE>
__global__ void k3_observe(const K3<T> a, T* out0, T* out1, T weight, int reset, int kl_first, int nzl) {
    // the cell layers between the local planes kl_first .. kl_first + nzl
    const size_t cells = (size_t)a.Nx * a.Ny * nzl;
    const size_t sz = (size_t)a.ny1 * a.px;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < cells; t += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(t % a.Nx), j = (int)((t / a.Nx) % a.Ny), kl = kl_first + (int)(t / ((size_t)a.Nx * a.Ny));
        const size_t id = ((size_t)kl * a.ny1 + j) * a.px + i;
        const bool wall = i + 1 >= a.Nx;          // column Nx of Ey, Ez is the unstored zero wall
        const T sx = load_e1<T, LEAN>(a, 0, id) + load_e1<T, LEAN>(a, 0, id + a.px) + load_e1<T, LEAN>(a, 0, id + sz) + load_e1<T, LEAN>(a, 0, id + sz + a.px);
        const T sy = load_e1<T, LEAN>(a, 1, id) + (wall ? (T)0 : load_e1<T, LEAN>(a, 1, id + 1)) + load_e1<T, LEAN>(a, 1, id + sz) + (wall ? (T)0 : load_e1<T, LEAN>(a, 1, id + sz + 1));
        const T s_z = load_e1<T, LEAN>(a, 2, id) + (wall ? (T)0 : load_e1<T, LEAN>(a, 2, id + 1)) + load_e1<T, LEAN>(a, 2, id + a.px) + (wall ? (T)0 : load_e1<T, LEAN>(a, 2, id + a.px + 1));
        if (MODE == 0) {
            const T e_sq = (T)0.0625 * (sx * sx + sy * sy + s_z * s_z);
            const T old = reset ? (T)0 : out0[id];
            out0[id] = old + e_sq * weight;
        } else {
            const T ax = (T)0.25 * sx, ay = (T)0.25 * sy, az = (T)0.25 * s_z;
            out0[t] = ax;
            out1[t] = ax * ax + ay * ay + az * az;
        }
    }
}

// The window array without its padding: [owned layers][Ny][Nx], x fastest.
template <typename T>
__global__ void k3_unpad(const T* __restrict__ padded, T* __restrict__ dense, int Nx, int Ny, int kl_first, int nzl, int px, int ny1) {
    const size_t cells = (size_t)Nx * Ny * nzl;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < cells; t += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(t % Nx), j = (int)((t / Nx) % Ny), kl = kl_first + (int)(t / ((size_t)Nx * Ny));
        dense[t] = padded[((size_t)kl * ny1 + j) * px + i];
    }
}

}  // namespace fdtd

using namespace fdtd;

// Address of `scratch` on device `dev`, which must be the calling thread's current device.  The name of a
// __device__ array is only a stand-in on the host side; the memory behind it is found once per device
// with hipGetSymbolAddress and that pointer is what kernels, memsets and copies are given.
static int static_scratch(int dev, double** out) {
    constexpr int kMaxDevices = 64;
    static std::mutex mu;
    static double* ptr[kMaxDevices] = {nullptr};
    if (dev < 0 || dev >= kMaxDevices) return fail(FDTD_ERR_ARG, "reduce_static: device ordinal %d outside [0,%d)", dev, kMaxDevices);
    std::lock_guard<std::mutex> lock(mu);
    if (!ptr[dev]) {
        int cur = -1;
        FDTD_HIP(hipGetDevice(&cur));
        if (cur != dev) return fail(FDTD_ERR_STATE, "reduce_static: device %d is current, the engine lives on %d", cur, dev);
        void* p = nullptr;
        FDTD_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(FDTD_ERR_HIP, "reduce_static: no address for the scratch array on device %d", dev);
        ptr[dev] = static_cast<double*>(p);
    }
    *out = ptr[dev];
    return FDTD_OK;
}

// Scratch memory of a two-step launch with CPML: 24 row segments (12 terms x 2 planes in flight) of one wave's
// columns per wave.  The kernel forms a wave's offset into it in 32 bits, so the whole piece must stay below 2^32
// elements (a 1024^3 grid at the default 128 planes per workgroup needs 0.35e9; 8 planes per workgroup would need 4.5e9).
extern "C" int fdtd3d_pair_scratch_fits(long long workgroups, int waves_per_group, int precision_bits) {
    if (workgroups < 0 || waves_per_group < 1 || (precision_bits != 32 && precision_bits != 64)) return 0;
    const unsigned long long tw = (unsigned long long)kLanes * (precision_bits == 32 ? 2 : 1);      // columns a wave computes
    const unsigned long long per_group = (unsigned long long)waves_per_group * 24ull * tw;
    return (unsigned long long)workgroups <= (((1ull << 32) - 1) / per_group) ? 1 : 0;
}

struct fdtd3d_engine {
    virtual ~fdtd3d_engine() {}
    virtual size_t array_size(int id) = 0;
    virtual int set_array(int id, const double* host, size_t n) = 0;
    virtual int get_array(int id, double* host, size_t n) = 0;
    virtual int fill_materials(uint64_t seed, double elo, double ehi, double mlo, double mhi) = 0;
    virtual int fill_reference_materials() = 0;
    virtual int fill_array(int id, uint64_t seed, double lo, double hi) = 0;
    virtual int half_e(int n, int part, hipStream_t s) = 0;
    virtual int half_h(int part, hipStream_t s) = 0;
    virtual int step(int n, hipStream_t s) = 0;
    virtual int fused_part(int n, int part, hipStream_t s) = 0;
    virtual int step_pair(int n, hipStream_t s) = 0;
    virtual int step_pair_part(int n, int part, hipStream_t s) = 0;
    virtual int pair_shape(int chunk, int rows) = 0;
    virtual bool pair_auto() = 0;
    virtual bool pair_capable() = 0;
    virtual void pair_give_up() = 0;
    virtual void pair_info(int out[4]) = 0;
    virtual void pair_layout(int out[4]) = 0;
    virtual int parity() = 0;
    virtual int fused_top(hipStream_t s) = 0;
    virtual int gaps(double* gib, int cap) = 0;
    virtual int probe_add(int id, int i, int j, int k) = 0;
    virtual int probe_record(hipStream_t s) = 0;
    virtual int probe_read(double* out, size_t cap, int* n_samples, int* n_probes) = 0;
    virtual int probe_reset(int drop_probes) = 0;
    virtual int probe_count() = 0;
    virtual int pair_peek_probes(int n, hipStream_t s) = 0;
    virtual void probe_unrecord() = 0;
    virtual int halo(int which, int comp, void** p, size_t* bytes, int at_parity) = 0;
    virtual int reduce(int id, double* sumsq, uint64_t* checksum, hipStream_t s) = 0;
    virtual int reduce_static(const int* ids, int n_ids, double* sumsq, uint64_t* checksum, hipStream_t s) = 0;
    virtual int reduce_range(int id, int k_lo, int k_hi, double* sumsq, uint64_t* checksum, hipStream_t s) = 0;
    virtual int planes(int id, int k, int count, void** p, size_t* nbytes) = 0;
    virtual size_t device_bytes() = 0;
    virtual int tune(int chunk, int remap, int rows) = 0;
    virtual void launch_shape(int out[6]) = 0;
    virtual int grant_access(int peer) = 0;
    virtual double spread_gap() = 0;
    virtual int observe_accumulate(double weight, int reset, hipStream_t s) = 0;
    virtual int observe_rms(double* host, size_t n) = 0;
    virtual int observe_keep_window() = 0;
    virtual int observe_rms_window(double* host, size_t n) = 0;
    virtual int step_observed(int n0, int count, double weight, int window, hipStream_t s) = 0;
    virtual int observe_set_range(int k_lo, int k_hi) = 0;
    virtual int pair_part_observed(int n, int part, double weight, int window, int defer_top, hipStream_t s) = 0;
    virtual int observe_finish_pair(int n, double weight, int window, int defer_top, hipStream_t s) = 0;
    virtual int observe_frame(double* ex_center, double* e_mag, size_t n, hipStream_t s) = 0;
    fdtd3d_desc d;
    KernelTimer timer;
    int device = -1;                 // HIP device the state lives on
};

namespace {

template <typename T>
struct Engine3 : fdtd3d_engine {
    static constexpr int V = VecWidth<T>::V;
    int px = 0, ny1 = 0, nzl = 0, nzp = 0, zoff = 0;
    int np[3][2];
    int xwn = 0, xwp = 0, xp0 = 0;
    bool last_rank = false, full = true, fused = false;
    T* alt[18] = {nullptr};           // fused stepping: the other D and B buffers of the ping-pong
    T* phE2[3][3][2];
    int wpbf = 8;                     // fused kernel: waves per workgroup (the last one is a halo row)
    int edge_planes = 1;              // fused kernel on a slab: planes of the EDGE and of the TOP launch (FDTD_EDGE_PLANES)
    int fvec = VecWidth<T>::V;        // fused kernel: elements per lane (V or V/2)
    size_t plane = 0, fsize = 0, bytes = 0;
    T* fld[18] = {nullptr};           // D E B H eps mu, three each (mu only if has_mu; E,H absent in LEAN)
    T* phE[3][3][2];
    T* phM[3][3][2];
    T* phM2[3][3][2] = {};            // second set of the B-side CPML state: only two-step launches need it
    bool phM2_ready = false;
    T* pair_scr = nullptr;
    size_t pair_scr_elems = 0;
    size_t phSize[2][3][3][2];
    T* tab_d = nullptr;
    T* xrow_d = nullptr;
    T* mag_d = nullptr;
    double* red_d = nullptr;          // reduction scratch: 1024 partial sums + checksum word
    T* obs_acc = nullptr;             // RMS window accumulator, [plane][row][column] like a field array
    T* obs_snap = nullptr;            // the last completed window (observe_keep_window), same shape
    bool obs_have_window = false;
    T* obs_tmp = nullptr;             // two frame buffers [owned layers][Ny][Nx] (allocated on first use)
    std::vector<double> tab_h;        // [3][2][6][kMaxLayers], fp64 master copy
    std::vector<T> wf;
    static constexpr int kRemapAuto = 65536;
    int chunk = 8, remap = kRemapAuto, wpb = 4;
    std::vector<void*> allocs;
    // arena backed through the virtual-memory API (spread placement, below): the reserved range and the physical
    // chunks mapped into it
    char* vm_base = nullptr;
    size_t vm_size = 0;
    std::vector<hipMemGenericAllocationHandle_t> vm_chunks;
    std::vector<size_t> big_starts;       // planning: arena offsets at which the large arrays begin
    double spread_gap_gib = 0;            // physical memory left out between two large arrays (0: not spread)

    void vm_release() {
        if (vm_base) {
            (void)hipMemUnmap(vm_base, vm_size);
            for (auto h : vm_chunks) (void)hipMemRelease(h);
            (void)hipMemAddressFree(vm_base, vm_size);
        }
        vm_base = nullptr; vm_size = 0; vm_chunks.clear();
    }

    ~Engine3() override {
        vm_release();
        for (void* p : allocs) (void)hipFree(p);
        if (stage_d) (void)hipFree(stage_d);
        probes.release();
        timer.release();
    }

    // ---- probe time series ----------------------------------------------------------------
    ProbeLog<T> probes;
    int probe_count() override { return (int)probes.specs.size(); }
    void probe_unrecord() override { if (probes.rows > 0) --probes.rows; }      // drops the newest row
    int probe_add(int id, int i, int j, int k) override {
        Map m;
        if (id < 0 || id >= 18 || !make_map(id, m)) return fail(FDTD_ERR_ARG, "probe_add: %d is not a field array", id);
        if (i < 0 || i >= m.n[0] || j < 0 || j >= m.n[1] || k < 0 || k >= m.n[2]) return fail(FDTD_ERR_ARG, "probe_add: cell (%d,%d,%d) outside array %d", i, j, k, id);
        bool owned;
        const long long off = dev_off(m, i, j, k, owned);
        if (off == -2) return fail(FDTD_ERR_ARG, "probe_add: column %d of array %d is the wall column, always zero", i, id);
        if (off < 0 || !owned) return fail(FDTD_ERR_ARG, "probe_add: layer %d is not owned by this slab [%d,%d)", k, d.z_lo, d.z_hi);
        ProbeSpec sp{id, -1, off, d.mu0};
        const int fam = id / 3, c = id % 3;
        if (!full && fam == 1) { sp.num = c; sp.den = 12 + c; }
        if (!full && fam == 3) { sp.num = 6 + c; sp.den = d.has_mu ? 15 + c : -2; }
        if (!fld[sp.num]) return fail(FDTD_ERR_ARG, "probe_add: array %d is not resident", id);
        return probes.add(sp);
    }
    int probe_record(hipStream_t s) override {
        if (probes.specs.empty()) return FDTD_OK;
        T* row = nullptr;
        if (int rc = probes.next_row(&row, (int)wf.size() + 8)) return rc;
        PtrTable<T, 18> t;
        for (int q = 0; q < 18; ++q) t.p[q] = fld[q];
        hipLaunchKernelGGL((k_probe<T, 18>), dim3(1), dim3(kMaxProbes), 0, s, t, probes.specs_d, (int)probes.specs.size(), row);
        FDTD_HIP(hipGetLastError());
        return FDTD_OK;
    }
    int probe_read(double* out, size_t cap, int* n_samples, int* n_probes) override { return probes.read(out, cap, n_samples, n_probes); }
    int probe_reset(int drop_probes) override { probes.reset(drop_probes != 0); return FDTD_OK; }

    // All arrays of the time loop are carved from one device allocation (4 KB granules), so that the
    // whole state can be placed -- and re-placed -- in device memory as one piece.  `planning` sums the
    // sizes first.
    char* arena = nullptr;
    size_t arena_cap = 0, arena_used = 0;
    bool planning = false;
    static constexpr size_t kGranule = 4096;
    size_t plain_gap = 0;                // while set: every large stand-alone allocation is preceded by a throw-away one of this size
    std::vector<void*> throwaway;
    int dalloc(void** p, size_t nbytes, int fill_byte_pattern_zero = 1) {
        if (nbytes == 0) { *p = nullptr; return FDTD_OK; }
        const size_t span = (nbytes + kGranule - 1) / kGranule * kGranule;
        const bool big = nbytes >= ((size_t)256 << 20);
        if (planning) { if (big) big_starts.push_back(arena_used); arena_used += span; *p = nullptr; return FDTD_OK; }
        if (arena && arena_used + span <= arena_cap) {
            *p = arena + arena_used;
            arena_used += span;
            bytes += nbytes;
        } else {                            // arrays created after set-up (observables) and the arrays of a slab engine stand alone
            if (plain_gap && big) {         // (see "Placement" in init)
                void* g = nullptr;
                if (hipMalloc(&g, plain_gap) == hipSuccess) { throwaway.push_back(g); gap_log.push_back((double)plain_gap / (double)((size_t)1 << 30)); }
                else { (void)hipGetLastError(); gap_log.push_back(0.0); }   // no room: no gap here
            }
            *p = nullptr;
            hipError_t got = hipMalloc(p, nbytes);
            if (got != hipSuccess && !throwaway.empty()) {
                // the gaps have taken what the arrays need: give them back at once and go on without
                (void)hipGetLastError();
                for (void* q : throwaway) (void)hipFree(q);
                throwaway.clear();
                plain_gap = 0;
                got = hipMalloc(p, nbytes);
            }
            FDTD_HIP(got);
            allocs.push_back(*p);
            bytes += nbytes;
        }
        if (fill_byte_pattern_zero) FDTD_HIP(hipMemset(*p, 0, nbytes));
        return FDTD_OK;
    }

    // GiB of physical memory left out in front of every large array but the first, in allocation order (empty: the
    // engine's memory is one plain allocation): what a run that streams slower than expected should look at first
    std::vector<double> gap_log;
    int gaps(double* gib, int cap) override {
        for (int q = 0; q < cap && q < (int)gap_log.size(); ++q) gib[q] = gap_log[q];
        return (int)gap_log.size();
    }

    // Spread placement.  The step kernels stream from all large arrays at once, and how fast that goes depends on how
    // far apart those arrays lie in PHYSICAL device memory: next to one another (what one big allocation gives) is the
    // slowest arrangement, a few GiB of other memory between two arrays the fastest (measured: 17.8 ms per 1024^3 step
    // with no gaps, 16.8 with 3 GiB, 16.65 with 7 GiB; tools/stream_spread.hip shows the same for a plain 18-read +
    // 6-write march).  The arena is therefore a reserved virtual range backed by 128 MiB physical chunks that are
    // created one after the other -- the driver hands physical memory out in order -- and in front of every large array
    // `gap` bytes worth of chunks are created too and released again as soon as the whole arena is mapped: the virtual
    // layout stays compact, the process only holds the memory of the arrays, and the arrays lie `gap` apart physically.
    // Returns FDTD_OK with vm_base == nullptr if the device cannot do it (the caller falls back to one allocation).
    static constexpr size_t kVmChunk = (size_t)128 << 20;
    int place_spread(size_t gap) {
        constexpr size_t kChunk = kVmChunk;
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = device;
        const size_t total = (arena_cap + kChunk - 1) / kChunk * kChunk;
        char* base = nullptr;
        if (hipMemAddressReserve((void**)&base, total, kChunk, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); return FDTD_OK; }
        vm_base = base; vm_size = total;
        std::vector<hipMemGenericAllocationHandle_t> dropped;
        const size_t gap_chunks = gap / kChunk;
        size_t next_big = 1;                                   // nothing is left out in front of the first array
        bool ok = true;
        size_t mapped = 0;
        for (size_t off = 0; off < total && ok; off += kChunk) {
            // a chunk in which the next large array begins: leave `gap` of physical memory out before it
            while (next_big < big_starts.size() && big_starts[next_big] < off + kChunk) {
                size_t made = 0;
                for (size_t g = 0; g < gap_chunks && ok; ++g) {
                    hipMemGenericAllocationHandle_t h;
                    if (hipMemCreate(&h, kChunk, &prop, 0) == hipSuccess) { dropped.push_back(h); ++made; } else { (void)hipGetLastError(); g = gap_chunks; }   // out of memory: smaller gaps from here on
                }
                gap_log.push_back((double)(made * kChunk) / (double)((size_t)1 << 30));
                ++next_big;
            }
            hipMemGenericAllocationHandle_t h;
            if (hipMemCreate(&h, kChunk, &prop, 0) != hipSuccess) { (void)hipGetLastError(); ok = false; break; }
            vm_chunks.push_back(h);
            if (hipMemMap(base + off, kChunk, 0, h, 0) != hipSuccess) { (void)hipGetLastError(); ok = false; break; }
            mapped = off + kChunk;
        }
        for (auto h : dropped) (void)hipMemRelease(h);
        if (ok) {
            hipMemAccessDesc acc = {};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            ok = hipMemSetAccess(base, total, &acc, 1) == hipSuccess;
            if (!ok) (void)hipGetLastError();
        }
        if (!ok) {
            if (mapped) (void)hipMemUnmap(base, mapped);
            for (auto h : vm_chunks) (void)hipMemRelease(h);
            (void)hipMemAddressFree(base, total);
            vm_base = nullptr; vm_size = 0; vm_chunks.clear(); gap_log.clear();
            return FDTD_OK;
        }
        spread_gap_gib = gap_log.empty() ? 0.0 : *std::min_element(gap_log.begin(), gap_log.end());
        return FDTD_OK;
    }
    // Lets device `peer` read and write the arena (copies between the devices of neighbouring slabs).
    int grant_access(int peer) override {
        if (!vm_base || peer == device) return FDTD_OK;
        hipMemAccessDesc acc = {};
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = peer;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        FDTD_HIP(hipMemSetAccess(vm_base, vm_size, &acc, 1));
        return FDTD_OK;
    }
    double spread_gap() override { return spread_gap_gib; }

    int allocate_all() {
        int rc;
        for (int f = 0; f < 6; ++f) for (int c = 0; c < 3; ++c) {
            const int id = f * 3 + c;
            fld[id] = nullptr;
            if (f == 5 && !d.has_mu) continue;
            if (!full && (f == 1 || f == 3)) continue;
            // one row of slack behind every array: vector accesses of the last row may look one pack ahead
            if ((rc = dalloc((void**)&fld[id], (fsize + px) * sizeof(T)))) return rc;
            if (f >= 4 && !planning) { if ((rc = fill_const(fld[id], fsize + px, (T)1))) return rc; }
            if (fused && (f == 0 || f == 2)) { if ((rc = dalloc((void**)&alt[id], (fsize + px) * sizeof(T)))) return rc; }
        }
        for (int m = 0; m < 2; ++m) for (int c = 0; c < 3; ++c) for (int a = 0; a < 3; ++a) for (int s = 0; s < 2; ++s) {
            T** slot = m ? &phM[c][a][s] : &phE[c][a][s];
            *slot = nullptr;
            phSize[m][c][a][s] = 0;
            if (a == c || np[a][s] == 0) continue;
            size_t n = 0;
            if (a == 2) n = (size_t)np[2][s] * plane;
            else if (a == 1) n = (size_t)nzp * np[1][s] * px;
            else n = (size_t)nzp * ny1 * (s ? xwp : xwn);
            phSize[m][c][a][s] = n;
            if ((rc = dalloc((void**)slot, (n + px) * sizeof(T)))) return rc;
            if (!m) { phE2[c][a][s] = nullptr; if (fused && (rc = dalloc((void**)&phE2[c][a][s], (n + px) * sizeof(T)))) return rc; }
        }
        if ((rc = dalloc((void**)&tab_d, (size_t)3 * 2 * 6 * kMaxLayers * sizeof(T)))) return rc;
        if ((rc = dalloc((void**)&xrow_d, (size_t)6 * px * sizeof(T)))) return rc;
        if ((rc = dalloc((void**)&mag_d, plane * sizeof(T)))) return rc;
        if ((rc = dalloc((void**)&red_d, 1025 * sizeof(double)))) return rc;
        return FDTD_OK;
    }

    int init(const fdtd3d_desc& dd) {
        d = dd;
        for (int a = 0; a < 3; ++a) for (int s = 0; s < 2; ++s) { np[a][s] = d.n_pml[a * 2 + s]; }
        px = round_up(d.Nx, kRowAlign);
        ny1 = d.Ny + 1;
        nzl = d.z_hi - d.z_lo;
        nzp = nzl + 2;
        zoff = d.z_lo - 1;
        last_rank = d.z_hi == d.Nz;
        full = d.mode == FDTD3D_MODE_FULL;
        fused = d.mode == FDTD3D_MODE_FUSED;
        plane = (size_t)ny1 * px;
        fsize = plane * nzp;
        xwn = np[0][0] > 0 ? round_up(np[0][0] + 1, 4) : 0;
        xp0 = (d.Nx - np[0][1]) / 4 * 4;
        xwp = np[0][1] > 0 ? round_up(d.Nx - xp0, 4) : 0;
        if (const char* e = getenv("FDTD_FUSED_VEC")) { int v = atoi(e); if (v == V || v == V / 2) fvec = v; }
        if (fused) {
            // The closing plane of a chunk is computed twice, so long chunks pay off -- once there are enough
            // workgroups to fill 256 CUs several times over.  Small grids (the reference's own 200^3) get shorter
            // chunks, down to 4 planes.  Workgroups have 8 waves (7 rows + the halo row): two of them share a CU and
            // run out of step with each other, which keeps memory busy while one of them sits at its per-plane barrier
            // -- since the kernel waits for memory only once per plane that is worth more than the halo rows the
            // 16-wave shape saves (1024^3: 15.3 against 15.8 ms; 512^3: 1.56 against 1.65; fdtd3d_tune selects 16).
            const int per_chunk = (px / (kLanes * fvec)) * ((ny1 + 14) / 15);
            const int chunks_wanted = (1400 + per_chunk - 1) / per_chunk;
            chunk = std::max(4, std::min(24, nzl / std::max(1, chunks_wanted)));
        }
        if (const char* e = getenv("FDTD_EDGE_PLANES")) { int c = atoi(e); if (c >= 1 && c <= 64) edge_planes = c; }
        if (const char* e = getenv("FDTD_Z_CHUNK")) { int c = atoi(e); if (c >= 1 && c <= 4096) chunk = c; }
        if (const char* e = getenv("FDTD_XCD_REMAP")) { int v = atoi(e); if (v >= 0 && v <= 65536) remap = v; }
        if (const char* e = getenv("FDTD_ROWS_PER_GROUP")) { int w = atoi(e); if (w >= 1 && w <= 4) wpb = w; }

        int rc;
        planning = true; arena_used = 0; big_starts.clear();
        if ((rc = allocate_all())) return rc;
        planning = false;
        arena_cap = arena_used;
        // Placement.  The step kernels stream from all field arrays at once, and how fast that goes depends on how far
        // apart those arrays lie in PHYSICAL device memory (place_spread).  Engines of 2 GiB or more therefore lay their
        // large arrays up to 10 GiB apart, less where the device does not have that much free: an engine that holds a
        // whole grid through the virtual-memory API, a slab engine -- whose planes travel between devices -- with
        // ordinary allocations and throw-away gaps.  Everything else, and any engine for which no gap of a useful size
        // is left, is one plain allocation.  (A 384^3 grid with CPML steps 6-7 % faster on spread arrays, 512^3 without
        // CPML 12 %; 256^3 and below do not notice; only gaps of about 10 GiB do it, 1-4 GiB change nothing.)
        // FDTD_SPREAD=0: never; FDTD_SPREAD=1: the virtual-memory way for every engine; FDTD_SPREAD_GAP_MB: the gap.
        const bool whole = d.z_lo == 0 && d.z_hi == d.Nz;
        int spread = arena_cap >= ((size_t)2 << 30) ? (whole ? 2 : 1) : 0;       // 2: virtual-memory API, 1: throw-away gaps
        if (const char* e = getenv("FDTD_SPREAD")) spread = e[0] == '1' ? 2 : 0;
        if (spread) {
            size_t free_b = 0, total_b = 0;
            FDTD_HIP(hipMemGetInfo(&free_b, &total_b));
            // 10 GiB where there is room: on a box whose memory nothing has used yet, the two-step kernel runs the 1024^3
            // grid in 11.12 ms per step with 10 GiB between the arrays, 11.17 with 8, 11.20 with 7, 11.34-11.47 with 6,
            // 11.85 with 4 (the one-step kernel does not care beyond 6)
            size_t gap = (size_t)10 << 30;
            if (const char* e = getenv("FDTD_SPREAD_GAP_MB")) gap = (size_t)std::max(0ll, atoll(e)) << 20;
            const size_t reserve = spread == 2 ? (size_t)2 << 30 : std::max((size_t)6 << 30, total_b / 32);
            const size_t n_gaps = big_starts.size() > 1 ? big_starts.size() - 1 : 1;
            const size_t room = free_b > arena_cap + reserve ? free_b - arena_cap - reserve : 0;
            gap = std::min(gap, room / n_gaps);
            if (spread == 2 && gap >= kVmChunk) {
                if ((rc = place_spread(gap))) return rc;
                if (vm_base) {
                    arena = vm_base; arena_used = 0; bytes = 0;
                    if ((rc = allocate_all())) return rc;
                }
            } else if (spread == 1 && gap >= ((size_t)1 << 30)) {
                // every array an allocation of its own, each large one preceded by a throw-away allocation of the gap, all
                // of which are freed once the arrays exist (the driver hands memory out in order); if an array does not
                // fit any more, dalloc gives the gaps back at once and the rest is allocated without
                struct Guard { std::vector<void*>& v; size_t& g; ~Guard() { for (void* q : v) (void)hipFree(q); v.clear(); g = 0; } } guard{throwaway, plain_gap};
                plain_gap = gap;
                arena = nullptr; arena_cap = 0; arena_used = 0; bytes = 0;
                if ((rc = allocate_all())) return rc;
                spread_gap_gib = gap_log.empty() ? 0.0 : *std::min_element(gap_log.begin(), gap_log.end());
            }
        }
        if (!vm_base && allocs.empty()) {          // one plain allocation
            arena_used = 0; bytes = 0;
            FDTD_HIP(hipMalloc((void**)&arena, arena_cap));
            allocs.push_back(arena);
            if ((rc = allocate_all())) return rc;
        }
        flips = 0;
        tab_h.assign((size_t)3 * 2 * 6 * kMaxLayers, 0.0);
        wf.assign(d.n_wf > 0 ? d.n_wf : 0, (T)0);
        return upload_tables();
    }

    int fill_const(T* p, size_t n, T v) {
        hipLaunchKernelGGL(k_fill_const<T>, dim3(2048), dim3(256), 0, 0, p, n, v);
        FDTD_HIP(hipGetLastError());
        return FDTD_OK;
    }

    size_t device_bytes() override { return bytes; }
    int tune(int c, int r, int w) override {
        if (c >= 1 && c <= 4096) chunk = c;
        if (r >= 0 && r <= 65536) remap = r;
        if (w >= 1 && w <= 4) wpb = w;
        if (fused && w >= 2 && w <= 16) wpbf = w;
        return FDTD_OK;
    }

    // what the step kernels are launched with: planes per workgroup, rows per workgroup, tile order,
    // elements per lane, planes of the EDGE/TOP launches, row pitch
    void launch_shape(int out[6]) override {
        out[0] = chunk;
        out[1] = fused ? wpbf : wpb;
        out[2] = remap == kRemapAuto ? (fused ? std::max(2, px / (kLanes * fvec)) : (full ? 4 : 1) * std::max(2, px / (kLanes * V))) : remap;
        out[3] = fused ? fvec : V;
        out[4] = edge_planes;
        out[5] = px;
    }

    // ---- canonical <-> device index maps ------------------------------------------------
    struct Map {
        int n[3] = {0, 0, 0};     // canonical extents
        T* dev = nullptr;
        T* dev2 = nullptr;        // second buffer that must receive the same upload (fused ping-pong)
        size_t dev_elems = 0;
        int kind = -1;            // 0 field, 1 phi z, 2 phi y, 3 phi x, 4 mag
        bool mag = false;
        int c = 0, a = 0, s = 0;
        size_t canon_size() const { return (size_t)n[0] * n[1] * n[2]; }
    };

    bool make_map(int id, Map& m) {
        const int N[3] = {d.Nx, d.Ny, d.Nz};
        if (id >= 0 && id < 18) {
            const int f = id / 3, c = id % 3;
            if (f == 5 && !d.has_mu) return false;
            const bool electric = (f == 0 || f == 1 || f == 4);
            for (int q = 0; q < 3; ++q) m.n[q] = electric ? N[q] + (q != c) : N[q] + (q == c);
            m.dev = fld[id]; m.dev2 = alt[id]; m.dev_elems = fsize; m.kind = 0; m.c = c;
            return true;
        }
        const bool isE = id >= FDTD3D_PHI_E && id < FDTD3D_PHI_E + 18, isM = id >= FDTD3D_PHI_M && id < FDTD3D_PHI_M + 18;
        if (isE || isM) {
            const int r = id - (isM ? FDTD3D_PHI_M : FDTD3D_PHI_E);
            m.s = r % 2; m.a = (r / 2) % 3; m.c = r / 6; m.mag = isM;
            if (m.a == m.c) return false;
            for (int q = 0; q < 3; ++q) m.n[q] = isM ? N[q] + (q == m.c) : N[q] + (q != m.c);
            m.n[m.a] = np[m.a][m.s];
            m.dev = isM ? phM[m.c][m.a][m.s] : phE[m.c][m.a][m.s];
            m.dev2 = isM ? (phM2_ready ? phM2[m.c][m.a][m.s] : nullptr) : (fused ? phE2[m.c][m.a][m.s] : nullptr);
            m.dev_elems = phSize[isM][m.c][m.a][m.s];
            m.kind = m.a == 2 ? 1 : (m.a == 1 ? 2 : 3);
            return true;
        }
        if (id == FDTD3D_JX_MAG) { m.n[0] = d.Nx; m.n[1] = d.Ny + 1; m.n[2] = 1; m.dev = mag_d; m.dev_elems = plane; m.kind = 4; return true; }
        return false;
    }
    // layer l of face (a,s) -> cell index along a (backward-difference terms start at 1)
    int layer_cell(bool mag, int a, int s, int l) const {
        const int N[3] = {d.Nx, d.Ny, d.Nz};
        if (s) return N[a] - np[a][1] + l;
        return mag ? l : 1 + l;
    }
    // device offset of canonical (p0,p1,p2), or -1 if this slab does not store it; owned tells
    // whether the slab is the one that reports it on download.
    long long dev_off(const Map& m, int p0, int p1, int p2, bool& owned) const {
        owned = false;
        auto own_k = [&](int k) { return (k >= d.z_lo && k < d.z_hi) || (last_rank && k == d.Nz); };
        switch (m.kind) {
            case 0: {
                if (p0 >= d.Nx) return -2;
                const int kl = p2 - zoff;
                if (kl < 0 || kl >= nzp) return -1;
                owned = own_k(p2);
                return ((long long)kl * ny1 + p1) * px + p0;
            }
            case 1: {       // z face: canonical (i, j, l), device [l][j][i]
                if (p0 >= d.Nx) return -2;
                owned = own_k(layer_cell(m.mag, 2, m.s, p2));
                return ((long long)p2 * ny1 + p1) * px + p0;
            }
            case 2: {       // y face: canonical (i, l, k), device [kl][l][i]
                if (p0 >= d.Nx) return -2;
                const int kl = p2 - zoff;
                if (kl < 0 || kl >= nzp) return -1;
                owned = own_k(p2);
                return ((long long)kl * np[1][m.s] + p1) * px + p0;
            }
            case 3: {       // x face: canonical (l, j, k), device [kl][j][column - strip origin]
                const int kl = p2 - zoff;
                if (kl < 0 || kl >= nzp) return -1;
                const int col = layer_cell(m.mag, 0, m.s, p0);
                if (col < 0 || col >= d.Nx) return -2;
                owned = own_k(p2);
                return m.s ? ((long long)kl * ny1 + p1) * xwp + (col - xp0) : ((long long)kl * ny1 + p1) * xwn + col;
            }
            case 4: owned = true; return (long long)p1 * px + p0;
        }
        return -1;
    }

    size_t array_size(int id) override {
        Map m;
        if (make_map(id, m)) return m.canon_size();
        if (id >= FDTD3D_TAB && id < FDTD3D_TAB + 36) return np[(id - FDTD3D_TAB) / 12][((id - FDTD3D_TAB) / 6) % 2];
        if (id == FDTD3D_JX_WF) return wf.size();
        return 0;
    }

    int upload_tables() {
        std::vector<T> t(tab_h.size());
        for (size_t q = 0; q < t.size(); ++q) t[q] = (T)tab_h[q];
        FDTD_HIP(hipMemcpy(tab_d, t.data(), t.size() * sizeof(T), hipMemcpyHostToDevice));
        // x-axis coefficient rows, one entry per column: CPML columns carry the tabulated (b, a, C),
        // plain columns (0, 0, dt/dx), columns the reference's sweeps skip (0, 0, 0).  Kinds 0-2 for
        // the backward-difference (D) terms, 3-5 for the forward-difference (B) terms.
        std::vector<T> xr((size_t)6 * px, (T)0);
        auto tb = [&](int s, int kind, int l) { return (T)tab_h[((size_t)(0 * 2 + s) * 6 + kind) * kMaxLayers + l]; };
        const T cmid = (T)(d.dt / d.dx);
        for (int i = 0; i < d.Nx; ++i) {
            int l;
            int z = zone_e(i, d.Nx, np[0][0], np[0][1], l);
            if (z == Z_NEG || z == Z_POS) for (int q = 0; q < 3; ++q) xr[(size_t)q * px + i] = tb(z == Z_POS, q, l);
            else if (z == Z_MID) xr[(size_t)2 * px + i] = cmid;
            z = zone_h(i, d.Nx, np[0][0], np[0][1], true, l);
            if (z == Z_NEG || z == Z_POS) for (int q = 3; q < 6; ++q) xr[(size_t)q * px + i] = tb(z == Z_POS, q, l);
            else if (z == Z_MID) xr[(size_t)5 * px + i] = cmid;
        }
        FDTD_HIP(hipMemcpy(xrow_d, xr.data(), xr.size() * sizeof(T), hipMemcpyHostToDevice));
        return FDTD_OK;
    }

    // ---- field arrays cross through a device-side transpose, 32 columns of the host array at a time ----
    double* stage_d = nullptr;
    size_t stage_cap = 0;
    int stage_reserve(size_t nbytes) {
        if (nbytes <= stage_cap) return FDTD_OK;
        if (stage_d) (void)hipFree(stage_d);
        stage_d = nullptr; stage_cap = 0;
        FDTD_HIP(hipMalloc((void**)&stage_d, nbytes));
        stage_cap = nbytes;
        return FDTD_OK;
    }
    void stage_trim() {          // a big staging buffer is only kept for the duration of one call
        if (stage_cap > ((size_t)64 << 20)) { (void)hipFree(stage_d); stage_d = nullptr; stage_cap = 0; }
    }
    ExchangeGeom exchange_geom(const Map& m, int i0, int ci) const {
        ExchangeGeom g;
        g.n1 = m.n[1]; g.n2 = m.n[2]; g.i0 = i0; g.ci = ci;
        g.Nx = d.Nx; g.px = px; g.ny1 = ny1; g.nzp = nzp; g.zoff = zoff;
        return g;
    }
    int set_field(const Map& m, const double* host) {
        const size_t per_i = (size_t)m.n[1] * m.n[2];
        if (int rc = stage_reserve(std::min<size_t>(32, m.n[0]) * per_i * sizeof(double))) return rc;
        const dim3 grid((m.n[2] + 31) / 32, m.n[1]);
        for (int i0 = 0; i0 < m.n[0] && i0 < d.Nx; i0 += 32) {
            const int ci = std::min(32, m.n[0] - i0);
            FDTD_HIP(hipMemcpy(stage_d, host + (size_t)i0 * per_i, (size_t)ci * per_i * sizeof(double), hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_scatter_field<T>, grid, dim3(256), 0, 0, stage_d, m.dev, exchange_geom(m, i0, ci));
            if (m.dev2) hipLaunchKernelGGL(k_scatter_field<T>, grid, dim3(256), 0, 0, stage_d, m.dev2, exchange_geom(m, i0, ci));
            FDTD_HIP(hipGetLastError());
        }
        FDTD_HIP(hipDeviceSynchronize());
        stage_trim();
        return FDTD_OK;
    }
    int get_field(const Map& m, const T* num, const T* den, T sdiv, double* host) {
        const size_t per_i = (size_t)m.n[1] * m.n[2];
        // layers this slab reports: its own, plus the closing plane k = Nz on the last slab
        const int k_lo = d.z_lo, k_hi = std::min(m.n[2], last_rank ? d.Nz + 1 : d.z_hi);
        if (int rc = stage_reserve(std::min<size_t>(32, m.n[0]) * per_i * sizeof(double))) return rc;
        std::vector<double> buf;
        const bool whole = k_lo == 0 && k_hi == m.n[2];
        if (!whole) buf.resize(std::min<size_t>(32, m.n[0]) * per_i);
        const dim3 grid((m.n[2] + 31) / 32, m.n[1]);
        for (int i0 = 0; i0 < m.n[0]; i0 += 32) {
            const int ci = std::min(32, m.n[0] - i0);
            double* dst = host + (size_t)i0 * per_i;
            if (i0 >= d.Nx) {               // the unstored wall column reads as zero
                for (size_t q = 0; q < (size_t)ci * per_i; ++q) dst[q] = 0.0;
                continue;
            }
            hipLaunchKernelGGL(k_gather_field<T>, grid, dim3(256), 0, 0, stage_d, num, den, sdiv, exchange_geom(m, i0, ci));
            FDTD_HIP(hipGetLastError());
            if (whole) {
                FDTD_HIP(hipMemcpy(dst, stage_d, (size_t)ci * per_i * sizeof(double), hipMemcpyDeviceToHost));
            } else {
                FDTD_HIP(hipMemcpy(buf.data(), stage_d, (size_t)ci * per_i * sizeof(double), hipMemcpyDeviceToHost));
                for (size_t row = 0; row < (size_t)ci * m.n[1]; ++row)
                    if (k_hi > k_lo) std::memcpy(dst + row * m.n[2] + k_lo, buf.data() + row * m.n[2] + k_lo, (size_t)(k_hi - k_lo) * sizeof(double));
            }
            for (int i = std::max(i0, d.Nx); i < i0 + ci; ++i)      // wall column inside a mixed chunk
                for (size_t q = 0; q < per_i; ++q) host[(size_t)i * per_i + q] = 0.0;
        }
        stage_trim();
        return FDTD_OK;
    }

    int set_array(int id, const double* host, size_t n) override {
        if (n != array_size(id) || (n && !host)) return fail(FDTD_ERR_ARG, "set_array(%d): expected %zu values, got %zu", id, array_size(id), n);
        if (id >= FDTD3D_TAB && id < FDTD3D_TAB + 36) {
            const int r = id - FDTD3D_TAB;
            for (size_t q = 0; q < n; ++q) tab_h[(size_t)r * kMaxLayers + q] = host[q];
            return upload_tables();
        }
        if (id == FDTD3D_JX_WF) { for (size_t q = 0; q < n; ++q) wf[q] = (T)host[q]; return FDTD_OK; }
        Map m;
        if (!make_map(id, m)) return fail(FDTD_ERR_ARG, "set_array: unknown array id %d", id);
        if (n == 0) return FDTD_OK;
        if (!m.dev) return fail(FDTD_ERR_STATE, "set_array(%d): array is not resident in this storage mode", id);
        if (m.kind == 0) return set_field(m, host);
        std::vector<T> st(m.dev_elems);
        FDTD_HIP(hipMemcpy(st.data(), m.dev, st.size() * sizeof(T), hipMemcpyDeviceToHost));
        size_t o = 0;
        for (int p0 = 0; p0 < m.n[0]; ++p0) for (int p1 = 0; p1 < m.n[1]; ++p1) for (int p2 = 0; p2 < m.n[2]; ++p2, ++o) {
            bool owned;
            const long long off = dev_off(m, p0, p1, p2, owned);
            if (off >= 0) st[(size_t)off] = (T)host[o];
        }
        FDTD_HIP(hipMemcpy(m.dev, st.data(), st.size() * sizeof(T), hipMemcpyHostToDevice));
        if (m.dev2) FDTD_HIP(hipMemcpy(m.dev2, st.data(), st.size() * sizeof(T), hipMemcpyHostToDevice));
        return FDTD_OK;
    }

    int get_array(int id, double* host, size_t n) override {
        if (n != array_size(id) || (n && !host)) return fail(FDTD_ERR_ARG, "get_array(%d): expected %zu values, got %zu", id, array_size(id), n);
        if (id >= FDTD3D_TAB && id < FDTD3D_TAB + 36) {
            for (size_t q = 0; q < n; ++q) host[q] = tab_h[(size_t)(id - FDTD3D_TAB) * kMaxLayers + q];
            return FDTD_OK;
        }
        if (id == FDTD3D_JX_WF) { for (size_t q = 0; q < n; ++q) host[q] = (double)wf[q]; return FDTD_OK; }
        Map m;
        if (!make_map(id, m)) return fail(FDTD_ERR_ARG, "get_array: unknown array id %d", id);
        if (n == 0) return FDTD_OK;
        FDTD_HIP(hipDeviceSynchronize());
        const bool derived = !full && m.kind == 0 && (id / 3 == 1 || id / 3 == 3);
        if (m.kind == 0) {
            if (!derived) {
                if (!m.dev) return fail(FDTD_ERR_STATE, "get_array(%d): array is not resident in this storage mode", id);
                return get_field(m, m.dev, nullptr, (T)0, host);
            }
            // lean storage: E = D/eps and H = B/mu are formed on the way out with the division the kernels use
            const int c = id % 3, num = id / 3 == 1 ? c : 6 + c, den = id / 3 == 1 ? 12 + c : 15 + c;
            return get_field(m, fld[num], fld[den], fld[den] ? (T)0 : (T)d.mu0, host);
        }
        // CPML auxiliaries and the source plane: small, transposed on the host
        if (!m.dev) return fail(FDTD_ERR_STATE, "get_array(%d): array is not resident in this storage mode", id);
        std::vector<T> st(m.dev_elems);
        FDTD_HIP(hipMemcpy(st.data(), m.dev, st.size() * sizeof(T), hipMemcpyDeviceToHost));
        size_t o = 0;
        for (int p0 = 0; p0 < m.n[0]; ++p0) for (int p1 = 0; p1 < m.n[1]; ++p1) for (int p2 = 0; p2 < m.n[2]; ++p2, ++o) {
            bool owned;
            const long long off = dev_off(m, p0, p1, p2, owned);
            if (off >= 0) { if (owned) host[o] = (double)st[(size_t)off]; }
            else if (off == -2) host[o] = 0.0;                          // unstored wall column
        }
        return FDTD_OK;
    }

    CompGeom geom(int id, bool owned_only) {
        CompGeom g;
        Map m; make_map(id, m);
        for (int q = 0; q < 3; ++q) g.n[q] = m.n[q];
        g.Nx = d.Nx; g.px = px; g.ny1 = ny1; g.zoff = zoff;
        g.kl_lo = owned_only ? 1 : 0;
        g.kl_hi = owned_only ? (last_rank ? nzp : nzp - 1) : nzp;
        return g;
    }

    int fill_materials(uint64_t seed, double elo, double ehi, double mlo, double mhi) override {
        for (int c = 0; c < 3; ++c) {
            hipLaunchKernelGGL(k_fill_hash<T>, dim3(2048), dim3(256), 0, 0, fld[12 + c], geom(12 + c, false), seed, (uint32_t)c, elo, ehi, 0, nzp);
            if (d.has_mu) hipLaunchKernelGGL(k_fill_hash<T>, dim3(2048), dim3(256), 0, 0, fld[15 + c], geom(15 + c, false), seed, (uint32_t)(3 + c), mlo, mhi, 0, nzp);
        }
        FDTD_HIP(hipGetLastError());
        FDTD_HIP(hipDeviceSynchronize());
        return FDTD_OK;
    }

    // any resident field array filled with the counter hash (measurements on a state without zeros)
    int fill_array(int id, uint64_t seed, double lo, double hi) override {
        if (id < 0 || id >= 18 || !fld[id]) return fail(FDTD_ERR_ARG, "fill_array: array %d is not a resident field", id);
        hipLaunchKernelGGL(k_fill_hash<T>, dim3(2048), dim3(256), 0, 0, fld[id], geom(id, false), seed, (uint32_t)(32 + id), lo, hi, 0, nzp);
        FDTD_HIP(hipGetLastError());
        FDTD_HIP(hipDeviceSynchronize());
        return FDTD_OK;
    }

    int fill_reference_materials() override {
        double eps0 = 0;
        fdtd_constants(nullptr, nullptr, &eps0, nullptr);
        const double n_glass = 1.47, n_air = 1.0;
        const double eps_glass = (n_glass * n_glass) * eps0, eps_air = (n_air * n_air) * eps0, radius = 2.0 * 1053e-9;
        for (int c = 0; c < 3; ++c)
            hipLaunchKernelGGL(k_fill_sphere<T>, dim3(4096), dim3(256), 0, 0, fld[12 + c], geom(12 + c, false), c, d.Nx, d.Ny, d.Nz,
                               d.dx, d.dy, d.dz, eps_glass, eps_air, radius, 0, nzp);
        FDTD_HIP(hipGetLastError());
        FDTD_HIP(hipDeviceSynchronize());
        return FDTD_OK;
    }

    int reduce(int id, double* sumsq, uint64_t* checksum, hipStream_t s) override {
        if (id < 0 || id >= 18 || !fld[id]) return fail(FDTD_ERR_ARG, "reduce: array %d is not a resident field", id);
        FDTD_HIP(hipMemsetAsync(red_d, 0, 1025 * sizeof(double), s));
        hipLaunchKernelGGL(k_reduce<T>, dim3(kReduceBlocks), dim3(256), 0, s, fld[id], geom(id, true), red_d, (unsigned long long*)(red_d + kReduceBlocks));
        FDTD_HIP(hipGetLastError());
        std::vector<double> h(1025);
        FDTD_HIP(hipMemcpyAsync(h.data(), red_d, 1025 * sizeof(double), hipMemcpyDeviceToHost, s));
        FDTD_HIP(hipStreamSynchronize(s));
        double acc = 0;
        for (int q = 0; q < 1024; ++q) acc += h[q];
        if (sumsq) *sumsq = acc;
        if (checksum) std::memcpy(checksum, &h[1024], sizeof(uint64_t));
        return FDTD_OK;
    }

    // the same over the global planes [k_lo, k_hi) only (a slab that stores more layers than it answers for)
    int reduce_range(int id, int k_lo, int k_hi, double* sumsq, uint64_t* checksum, hipStream_t s) override {
        if (id < 0 || id >= 18 || !fld[id]) return fail(FDTD_ERR_ARG, "reduce: array %d is not a resident field", id);
        CompGeom g = geom(id, true);
        g.kl_lo = std::max(g.kl_lo, k_lo - zoff);
        g.kl_hi = std::min(g.kl_hi, k_hi - zoff);
        if (g.kl_hi < g.kl_lo) g.kl_hi = g.kl_lo;
        FDTD_HIP(hipMemsetAsync(red_d, 0, 1025 * sizeof(double), s));
        hipLaunchKernelGGL(k_reduce<T>, dim3(kReduceBlocks), dim3(256), 0, s, fld[id], g, red_d, (unsigned long long*)(red_d + kReduceBlocks));
        FDTD_HIP(hipGetLastError());
        std::vector<double> h(1025);
        FDTD_HIP(hipMemcpyAsync(h.data(), red_d, 1025 * sizeof(double), hipMemcpyDeviceToHost, s));
        FDTD_HIP(hipStreamSynchronize(s));
        double acc = 0;
        for (int q = 0; q < 1024; ++q) acc += h[q];
        if (sumsq) *sumsq = acc;
        if (checksum) std::memcpy(checksum, &h[1024], sizeof(uint64_t));
        return FDTD_OK;
    }
    // `count` consecutive stored planes of a D / B array or of an x- or y-face CPML array, starting at global plane k,
    // in the buffers that are current now: pointer arithmetic only
    int planes(int id, int k, int count, void** p, size_t* nbytes) override {
        Map m;
        if (!make_map(id, m) || !m.dev || (m.kind != 0 && m.kind != 2 && m.kind != 3)) return fail(FDTD_ERR_ARG, "planes: array %d has no per-plane storage here", id);
        const int kl = k - zoff;
        if (count < 1 || kl < 0 || kl + count > nzp) return fail(FDTD_ERR_ARG, "planes: [%d, %d) is not stored by this slab", k, k + count);
        const size_t per = m.kind == 0 ? plane : (m.kind == 2 ? (size_t)np[1][m.s] * px : (size_t)ny1 * (m.s ? xwp : xwn));
        *p = m.dev + (size_t)kl * per;
        *nbytes = (size_t)count * per * sizeof(T);
        return FDTD_OK;
    }

    // The same reduction for up to kStaticMaxIds arrays in one call, with the partial sums in the file-scope
    // scratch array: no hipMalloc, no hipFree, no host container.  One stream synchronisation at the end.
    int reduce_static(const int* ids, int n_ids, double* sumsq, uint64_t* checksum, hipStream_t s) override {
        if (n_ids < 0 || (n_ids > 0 && !ids)) return fail(FDTD_ERR_ARG, "reduce_static: bad id list");
        if (n_ids > kStaticMaxIds) return fail(FDTD_ERR_ARG, "reduce_static: %d arrays asked for, the static scratch holds partial sums for at most %d", n_ids, kStaticMaxIds);
        for (int q = 0; q < n_ids; ++q)
            if (ids[q] < 0 || ids[q] >= 18 || !fld[ids[q]]) return fail(FDTD_ERR_ARG, "reduce_static: array %d is not a resident field", ids[q]);
        if (n_ids == 0) return FDTD_OK;
        double* sc = nullptr;
        if (int rc = static_scratch(device, &sc)) return rc;
        double* results = sc + (size_t)kStaticMaxIds * kStaticSlot;
        // the array keeps what the previous call left: the checksum words are accumulated into, so they start at zero
        FDTD_HIP(hipMemsetAsync(sc, 0, (size_t)n_ids * kStaticSlot * sizeof(double), s));
        for (int q = 0; q < n_ids; ++q) {
            double* slot = sc + (size_t)q * kStaticSlot;
            hipLaunchKernelGGL(k_reduce<T>, dim3(kReduceBlocks), dim3(256), 0, s, fld[ids[q]], geom(ids[q], true), slot, (unsigned long long*)(slot + kReduceBlocks));
        }
        hipLaunchKernelGGL(k_reduce_finish, dim3(1), dim3(64), 0, s, sc, results, n_ids);
        FDTD_HIP(hipGetLastError());
        double host[2 * kStaticMaxIds];
        FDTD_HIP(hipMemcpyAsync(host, results, (size_t)2 * n_ids * sizeof(double), hipMemcpyDeviceToHost, s));
        FDTD_HIP(hipStreamSynchronize(s));
        for (int q = 0; q < n_ids; ++q) {
            if (sumsq) sumsq[q] = host[2 * q];
            if (checksum) std::memcpy(&checksum[q], &host[2 * q + 1], sizeof(uint64_t));
        }
        return FDTD_OK;
    }

    // ---- observables ---------------------------------------------------------------------
    // the layers whose observables this engine reports (a slab that overlaps its neighbours stores more than it answers for)
    int ans_lo = -1, ans_hi = -1;
    int obs_lo() const { return ans_lo >= 0 ? ans_lo : d.z_lo; }
    int obs_hi() const { return ans_lo >= 0 ? ans_hi : d.z_hi; }
    int obs_kl() const { return 1 + obs_lo() - d.z_lo; }                 // local plane of the first reported layer
    size_t obs_cells() const { return (size_t)d.Nx * d.Ny * (obs_hi() - obs_lo()); }
    int observe_set_range(int k_lo, int k_hi) override {
        if (k_lo < d.z_lo || k_hi > d.z_hi || k_lo >= k_hi) return fail(FDTD_ERR_ARG, "observe_set_range: [%d,%d) is not inside the stored layers [%d,%d)", k_lo, k_hi, d.z_lo, d.z_hi);
        if (obs_tmp) return fail(FDTD_ERR_STATE, "observe_set_range: set the range before the first download");
        ans_lo = k_lo; ans_hi = k_hi;
        return FDTD_OK;
    }
    int obs_reserve(bool snap) {
        int rc;
        if (!obs_acc && (rc = dalloc((void**)&obs_acc, (fsize + px) * sizeof(T)))) return rc;
        if (snap && !obs_snap && (rc = dalloc((void**)&obs_snap, (fsize + px) * sizeof(T)))) return rc;
        return FDTD_OK;
    }
    int observe_accumulate(double weight, int reset, hipStream_t s) override {
        if (int rc = obs_reserve(false)) return rc;
        return accumulate_layers(weight, reset, 1, nzl, s);
    }
    // the window of the cell layers between the local planes kl_first .. kl_first + count advanced by one step
    int accumulate_layers(double weight, int reset, int kl_first, int count, hipStream_t s) {
        K3<T> a = args();
        if (full) hipLaunchKernelGGL((k3_observe<T, false, 0>), dim3(4096), dim3(256), 0, s, a, obs_acc, (T*)nullptr, (T)weight, reset, kl_first, count);
        else hipLaunchKernelGGL((k3_observe<T, true, 0>), dim3(4096), dim3(256), 0, s, a, obs_acc, (T*)nullptr, (T)weight, reset, kl_first, count);
        FDTD_HIP(hipGetLastError());
        return FDTD_OK;
    }
    // The window as it stands becomes "the last completed window" (observe_rms_window): the two arrays change roles,
    // nothing is copied.  What the accumulator holds afterwards is undefined: the next accumulation must open a new
    // window (reset), which is what follows a completed window anyway.
    int observe_keep_window() override {
        if (int rc = obs_reserve(true)) return rc;
        std::swap(obs_acc, obs_snap);
        obs_have_window = true;
        return FDTD_OK;
    }
    // scatter an [owned layers][Ny][Nx] device array into the global [Nz][Ny][Nx] host array
    int obs_download(const T* dev, double* host, bool take_sqrt) {
        std::vector<T> st(obs_cells());
        FDTD_HIP(hipDeviceSynchronize());
        FDTD_HIP(hipMemcpy(st.data(), dev, st.size() * sizeof(T), hipMemcpyDeviceToHost));
        double* dst = host + (size_t)obs_lo() * d.Ny * d.Nx;
        for (size_t q = 0; q < st.size(); ++q) dst[q] = take_sqrt ? std::sqrt((double)st[q]) : (double)st[q];
        return FDTD_OK;
    }
    int obs_download_window(const T* padded, double* host) {
        int rc;
        if (!obs_tmp && (rc = dalloc((void**)&obs_tmp, 2 * obs_cells() * sizeof(T)))) return rc;
        FDTD_HIP(hipDeviceSynchronize());
        hipLaunchKernelGGL(k3_unpad<T>, dim3(4096), dim3(256), 0, 0, padded, obs_tmp, d.Nx, d.Ny, obs_kl(), obs_hi() - obs_lo(), px, ny1);
        FDTD_HIP(hipGetLastError());
        return obs_download(obs_tmp, host, true);
    }
    int observe_rms(double* host, size_t n) override {
        if (n != (size_t)d.Nx * d.Ny * d.Nz || !host) return fail(FDTD_ERR_ARG, "observe_rms: expected %zu values", (size_t)d.Nx * d.Ny * d.Nz);
        if (!obs_acc) return fail(FDTD_ERR_STATE, "observe_rms: nothing accumulated yet");
        return obs_download_window(obs_acc, host);
    }
    int observe_rms_window(double* host, size_t n) override {
        if (n != (size_t)d.Nx * d.Ny * d.Nz || !host) return fail(FDTD_ERR_ARG, "observe_rms_window: expected %zu values", (size_t)d.Nx * d.Ny * d.Nz);
        if (!obs_have_window) return fail(FDTD_ERR_STATE, "observe_rms_window: no window has been completed yet");
        return obs_download_window(obs_snap, host);
    }
    // Steps n, n+1 of a SLAB in one launch (or its EDGE / INTERIOR part) with the window advanced by both, for drivers of
    // overlapping slabs (fdtd3d_pair_part).  `defer_top`: the slab has a neighbour above; the second step's contribution
    // of its top reported layer needs E one layer further up, which is exact only once the planes beyond the cut have
    // arrived, so the launch leaves it out and observe_finish_pair adds it.
    int pair_part_observed(int n, int part, double weight, int window, int defer_top, hipStream_t s) override {
        if (window < 1 || n < 0) return fail(FDTD_ERR_ARG, "pair_part_observed: bad window or step number");
        if (!probes.specs.empty()) return fail(FDTD_ERR_STATE, "two-step launches cannot sample probes after the first of their steps");
        if (int rc = obs_reserve(true)) return rc;
        const bool done1 = (n + 1) % window == 0;
        pair_acc_w = (T)weight;
        pair_acc_flags = (n % window == 0 ? 1 : 0) | (done1 ? 2 | 4 : 0);
        pair_acc_defer = defer_top ? obs_hi() - d.z_lo : -1;           // local plane of the top reported layer (plane = layer - z_lo + 1)
        const int rc = step_pair_part(n, part, s);
        pair_acc_flags = -1; pair_acc_defer = -1;
        if (rc) return rc;
        if (done1 && part != FDTD3D_PART_EDGE) obs_have_window = true;
        return FDTD_OK;
    }
    // After the planes beyond the cuts have arrived: the deferred contribution, and the change of roles of the two window
    // arrays if the launch's second step completed a window.
    int observe_finish_pair(int n, double weight, int window, int defer_top, hipStream_t s) override {
        if (window < 1 || n < 0) return fail(FDTD_ERR_ARG, "observe_finish_pair: bad window or step number");
        if (defer_top) if (int rc = accumulate_layers(weight, (n + 1) % window == 0, obs_hi() - d.z_lo, 1, s)) return rc;
        if ((n + 2) % window == 0) return observe_keep_window();
        return FDTD_OK;
    }
    // `count` steps from n0, each followed by the accumulation of FDTD3D::run (FDTD3D.cpp:1089-1097): a new window opens
    // at every step n with n % window == 0, and the window that step n completes ((n + 1) % window == 0) is kept for
    // observe_rms_window.  Where fdtd3d_step would take two steps per launch, the launch advances the window by both
    // of its steps as well (and, when its first step completes a window, keeps it on the way).
    int step_observed(int n0, int count, double weight, int window, hipStream_t s) override {
        if (window < 1) return fail(FDTD_ERR_ARG, "step_observed: the window must have at least one step");
        if (n0 < 0) return fail(FDTD_ERR_ARG, "step_observed: negative step number");
        if (int rc = obs_reserve(true)) return rc;
        const bool probed = !probes.specs.empty();
        for (int n = n0; n < n0 + count;) {
            if (n + 2 <= n0 + count && pair_auto()) {
                if (probed) if (int rc = pair_peek_probes(n, s)) return rc;
                const bool done1 = (n + 1) % window == 0;              // the first step completes a window: the second opens one
                pair_acc_w = (T)weight;
                pair_acc_flags = (n % window == 0 ? 1 : 0) | (done1 ? 2 | 4 : 0);
                const int rc = step_pair(n, s);
                pair_acc_flags = -1;
                if (rc == FDTD_OK) {
                    if (done1) obs_have_window = true;
                    if ((n + 2) % window == 0) if (int rc2 = observe_keep_window()) return rc2;
                    if (int rc2 = probe_record(s)) return rc2;
                    n += 2;
                    continue;
                }
                if (rc != FDTD_ERR_HIP) return rc;                      // (no device memory for the launch: single steps, as fdtd3d_step)
                if (probed) probe_unrecord();
                pair_give_up();
            }
            if (int rc = step(n, s)) return rc;
            if (int rc = observe_accumulate(weight, n % window == 0, s)) return rc;
            if ((n + 1) % window == 0) if (int rc = observe_keep_window()) return rc;
            if (int rc = probe_record(s)) return rc;
            ++n;
        }
        return FDTD_OK;
    }
    int observe_frame(double* ex_center, double* e_mag, size_t n, hipStream_t s) override {
        if (n != (size_t)d.Nx * d.Ny * d.Nz || !ex_center || !e_mag) return fail(FDTD_ERR_ARG, "observe_frame: expected %zu values per array", (size_t)d.Nx * d.Ny * d.Nz);
        int rc;
        if (!obs_tmp && (rc = dalloc((void**)&obs_tmp, 2 * obs_cells() * sizeof(T)))) return rc;
        K3<T> a = args();
        if (full) hipLaunchKernelGGL((k3_observe<T, false, 1>), dim3(4096), dim3(256), 0, s, a, obs_tmp, obs_tmp + obs_cells(), (T)0, 0, obs_kl(), obs_hi() - obs_lo());
        else hipLaunchKernelGGL((k3_observe<T, true, 1>), dim3(4096), dim3(256), 0, s, a, obs_tmp, obs_tmp + obs_cells(), (T)0, 0, obs_kl(), obs_hi() - obs_lo());
        FDTD_HIP(hipGetLastError());
        if ((rc = obs_download(obs_tmp, ex_center, false))) return rc;
        return obs_download(obs_tmp + obs_cells(), e_mag, true);
    }

    // ---- launches -----------------------------------------------------------------------
    K3<T> args() {
        K3<T> a;
        for (int c = 0; c < 3; ++c) {
            a.D[c] = fld[c]; a.E[c] = fld[3 + c]; a.B[c] = fld[6 + c]; a.H[c] = fld[9 + c];
            a.eps[c] = fld[12 + c]; a.mu[c] = fld[15 + c];
            for (int ax = 0; ax < 3; ++ax) for (int s = 0; s < 2; ++s) { a.phE[c][ax][s] = phE[c][ax][s]; a.phM[c][ax][s] = phM[c][ax][s]; }
        }
        a.tab = tab_d; a.xrow = xrow_d; a.mag = mag_d;
        a.Nx = d.Nx; a.Ny = d.Ny; a.Nz = d.Nz; a.px = px; a.ny1 = ny1; a.nzp = nzp; a.zoff = zoff;
        for (int ax = 0; ax < 3; ++ax) for (int s = 0; s < 2; ++s) a.np[ax][s] = np[ax][s];
        a.xwn = xwn; a.xwp = xwp; a.xp0 = xp0;
        a.wrap_l = 0; a.wrap_r0 = 0; a.peek = 0;
        a.acc = nullptr; a.acc_snap = nullptr; a.acc_w = (T)0; a.acc_flags = 0; a.acc_defer_kl = -1;
        a.kJ = d.k_Jx; a.src_on = 0; a.jx = (T)0;
        a.dt = (T)d.dt; a.dtd[0] = (T)(d.dt / d.dx); a.dtd[1] = (T)(d.dt / d.dy); a.dtd[2] = (T)(d.dt / d.dz); a.mu0 = (T)d.mu0;
        a.nbx = px / (kLanes * V); a.nby = (ny1 + wpb - 1) / wpb; a.wpb = wpb;
        a.chunk = chunk;
        // half-step kernels: runs of one row group's x tiles per XCD (measured at 1024^3: LEAN 21.4 ms against
        // 23.5 ms with contiguous ranges; with E and H resident too, four row groups per run are slightly better)
        a.remap = remap == kRemapAuto ? (full ? 4 : 1) * std::max(2, px / (kLanes * V)) : remap;
        return a;
    }
    // planes [kl0,kl1) of a part; E-half edge = lowest owned plane, H-half edge = highest owned
    // cell layer (the closing wall plane of the last slab counts as interior).
    bool part_range(bool eh, int part, int& kl0, int& kl1) const {
        const int lo = 1, hi = last_rank ? nzp : nzp - 1;   // owned planes [lo,hi)
        const int top_layer = nzl;                           // plane of cell layer z_hi-1
        if (part == FDTD3D_PART_ALL) { kl0 = lo; kl1 = hi; return true; }
        if (eh) {
            if (part == FDTD3D_PART_EDGE) { kl0 = lo; kl1 = lo + 1; return true; }
            if (part == FDTD3D_PART_INTERIOR) { kl0 = lo + 1; kl1 = hi; return true; }
        } else {
            if (part == FDTD3D_PART_EDGE) { kl0 = top_layer; kl1 = top_layer + 1; return true; }
        }
        return false;
    }
    template <typename Kern>
    int launch(Kern kern, K3<T> a, int kl0, int kl1, int kind, hipStream_t s, int part = 0) {
        if (kl1 <= kl0) return FDTD_OK;
        a.kl0 = kl0; a.kl1 = kl1;
        a.nbz = (kl1 - kl0 + a.chunk - 1) / a.chunk;
        const unsigned nblk = (unsigned)a.nbx * a.nby * a.nbz;
        timer.begin(kind, s, part);
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(kLanes * wpb), 0, s, a);
        timer.end(s);
        FDTD_HIP(hipGetLastError());
        return FDTD_OK;
    }
    int launch_e(const K3<T>& a, int kl0, int kl1, hipStream_t s, int part = 0) {
        if (full) return d.has_mu ? launch(k3_half_e<T, true, false>, a, kl0, kl1, 0, s, part) : launch(k3_half_e<T, false, false>, a, kl0, kl1, 0, s, part);
        return d.has_mu ? launch(k3_half_e<T, true, true>, a, kl0, kl1, 0, s, part) : launch(k3_half_e<T, false, true>, a, kl0, kl1, 0, s, part);
    }
    int launch_h(const K3<T>& a, int kl0, int kl1, hipStream_t s, int part = 0) {
        if (full) return d.has_mu ? launch(k3_half_h<T, true, false>, a, kl0, kl1, 1, s, part) : launch(k3_half_h<T, false, false>, a, kl0, kl1, 1, s, part);
        return d.has_mu ? launch(k3_half_h<T, true, true>, a, kl0, kl1, 1, s, part) : launch(k3_half_h<T, false, true>, a, kl0, kl1, 1, s, part);
    }
    int half_e(int n, int part, hipStream_t s) override {
        if (pair_phase != 0) return fail(FDTD_ERR_STATE, "half_e: a two-step launch is half done (EDGE without INTERIOR)");
        K3<T> a = args();
        a.src_on = (n >= 0 && n < (int)wf.size()) ? 1 : 0;
        a.jx = a.src_on ? a.dt * wf[n] : (T)0;      // product formed first, as FDTD3D.cpp:482
        int kl0, kl1;
        if (!part_range(true, part, kl0, kl1)) return fail(FDTD_ERR_ARG, "half_e: bad part %d", part);
        return launch_e(a, kl0, kl1, s, part);
    }
    int half_h(int part, hipStream_t s) override {
        if (pair_phase != 0) return fail(FDTD_ERR_STATE, "half_h: a two-step launch is half done (EDGE without INTERIOR)");
        K3<T> a = args();
        int kl0, kl1;
        if (part == FDTD3D_PART_INTERIOR) {
            // everything except the highest owned cell layer: the planes below it, and the closing plane above it
            int rc = launch_h(a, 1, nzl, s, part);
            if (rc || !last_rank) return rc;
            return launch_h(a, nzl + 1, nzp, s, part);
        }
        if (!part_range(false, part, kl0, kl1)) return fail(FDTD_ERR_ARG, "half_h: bad part %d", part);
        return launch_h(a, kl0, kl1, s, part);
    }

    template <int FV>
    void launch_fused(K3<T> a, hipStream_t s) {
        a.nbx = px / (kLanes * FV);
        const unsigned nblk = (unsigned)a.nbx * a.nby * a.nbz;
        const size_t lds = ((size_t)2 * wpbf * 3 + 7) * kLanes * FV * sizeof(T);       // E rows of two planes + the tile's x coefficients
        auto go = [&](auto kern) {
            // more than 64 KB of dynamic LDS has to be allowed per kernel
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, dim3(nblk), dim3(kLanes * wpbf), lds, s, a);
        };
        if (wpbf > 8) { if (d.has_mu) go(k3_fused<T, true, 16, FV>); else go(k3_fused<T, false, 16, FV>); }
        else { if (d.has_mu) go(k3_fused<T, true, 8, FV>); else go(k3_fused<T, false, 8, FV>); }
    }
    // One launch per time step: D and B cross HBM once.  On a slab the lowest chunk of planes can be
    // launched first (EDGE) so that its D planes travel while the rest (INTERIOR) is computed; the
    // buffers change roles when the first part of a step is launched.
    int phase = 0;                   // 0 between steps, 1 after EDGE, 2 after EDGE + TOP
    int flips = 0;
    int parity() override { return flips & 1; }
    int step(int n, hipStream_t s) override {
        if (fused) return fused_part(n, FDTD3D_PART_ALL, s);
        if (int rc = half_e(n, FDTD3D_PART_ALL, s)) return rc;
        return half_h(FDTD3D_PART_ALL, s);
    }
    // B of the top layer of a slab that has another one above it (no-op on the last slab)
    int fused_top(hipStream_t s) override {
        if (!fused) return fail(FDTD_ERR_STATE, "fused stepping needs an engine created with FDTD3D_MODE_FUSED");
        if (phase == 1) return fail(FDTD_ERR_STATE, "fused: the part holding the top layer (TOP or INTERIOR) has not been launched");
        return last_rank ? FDTD_OK : half_h(FDTD3D_PART_EDGE, s);
    }
    // arguments of a one-step fused launch for step n: reads the current buffers, writes the other set (`swapped`: the
    // roles were already exchanged when the step's first part was launched)
    K3<T> fused_args(int n, bool swapped) {
        K3<T> a = args();
        a.src_on = (n >= 0 && n < (int)wf.size()) ? 1 : 0;
        a.jx = a.src_on ? a.dt * wf[n] : (T)0;
        for (int c = 0; c < 3; ++c) {
            a.Dn[c] = alt[c]; a.Bn[c] = alt[6 + c];
            for (int ax = 0; ax < 3; ++ax) for (int sd = 0; sd < 2; ++sd) a.phEn[c][ax][sd] = phE2[c][ax][sd];
        }
        if (swapped)
            for (int c = 0; c < 3; ++c) {
                std::swap(a.D[c], a.Dn[c]); std::swap(a.B[c], a.Bn[c]);
                for (int ax = 0; ax < 3; ++ax) for (int sd = 0; sd < 2; ++sd) std::swap(a.phE[c][ax][sd], a.phEn[c][ax][sd]);
            }
        a.wpb = wpbf;
        // tile order of the fused kernel: the x tiles of one row group share an XCD (their edge columns meet
        // in its L2), successive row groups go round the XCDs, so that all XCDs stream from the same planes
        if (remap == kRemapAuto) a.remap = std::max(2, px / (kLanes * fvec));
        a.nby = (ny1 + (wpbf - 1) - 1) / (wpbf - 1);
        a.kl_stop = last_rank ? nzp : nzp - 1; a.copy_top = last_rank ? 0 : 1;
        return a;
    }
    int fused_go(K3<T>& a, int kl0, int kl1, int part, hipStream_t s) {
        a.kl0 = kl0; a.kl1 = kl1;
        a.nbz = (a.kl1 - a.kl0 + a.chunk - 1) / a.chunk;
        if (kl1 > kl0) {
            timer.begin(2, s, part);
            if (fvec == V) launch_fused<V>(a, s); else launch_fused<V / 2>(a, s);
            timer.end(s);
            FDTD_HIP(hipGetLastError());
        }
        return FDTD_OK;
    }
    // Probes on a two-step launch: the values after the FIRST of the two steps never reach memory, so they are formed
    // here, before the launch, by one-plane launches of the one-step kernel over the planes that hold probes.  These
    // read the current buffers and write the other set -- which the two-step launch then overwrites -- and leave the
    // B-side CPML state, which the one-step kernel keeps in place, untouched (`peek`).  The row is sampled from the
    // other set; the second step's row is an ordinary probe_record() after the launch.
    int pair_peek_probes(int n, hipStream_t s) override {
        if (probes.specs.empty()) return FDTD_OK;
        if (!fused || phase != 0 || pair_phase != 0) return fail(FDTD_ERR_STATE, "probes before a two-step launch: engine is in the middle of a step");
        std::vector<int> planes;
        for (const ProbeSpec& sp : probes.specs) {
            const int kl = (int)(sp.off / (long long)plane);
            if (std::find(planes.begin(), planes.end(), kl) == planes.end()) planes.push_back(kl);
        }
        K3<T> a = fused_args(n, false);
        a.peek = 1;
        for (int kl : planes) if (int rc = fused_go(a, kl, kl + 1, FDTD3D_PART_ALL, s)) return rc;
        T* row = nullptr;
        if (int rc = probes.next_row(&row, (int)wf.size() + 8)) return rc;
        PtrTable<T, 18> t;
        for (int q = 0; q < 18; ++q) t.p[q] = fld[q];
        for (int c = 0; c < 3; ++c) { t.p[c] = alt[c]; t.p[6 + c] = alt[6 + c]; }
        hipLaunchKernelGGL((k_probe<T, 18>), dim3(1), dim3(kMaxProbes), 0, s, t, probes.specs_d, (int)probes.specs.size(), row);
        FDTD_HIP(hipGetLastError());
        return FDTD_OK;
    }
    int fused_part(int n, int part, hipStream_t s) override {
        if (!fused) return fail(FDTD_ERR_STATE, "fused stepping needs an engine created with FDTD3D_MODE_FUSED");
        if (pair_phase != 0) return fail(FDTD_ERR_STATE, "fused: a two-step launch is half done (EDGE without INTERIOR)");
        // EDGE only has to produce layer z_lo early, TOP only the top layer: these launches have too few
        // workgroups to fill the chip, so they take as few planes as possible and leave the rest to INTERIOR
        const int lo = 1, hi = last_rank ? nzp : nzp - 1, few = std::min(chunk, edge_planes);
        const int cut = std::min(lo + few, hi), cut2 = std::max(cut, hi - few);
        int kl0, kl1;
        if (part == FDTD3D_PART_ALL) { kl0 = lo; kl1 = hi; }
        else if (part == FDTD3D_PART_EDGE) { kl0 = lo; kl1 = cut; }
        else if (part == FDTD3D_PART_TOP) { kl0 = cut2; kl1 = hi; }
        else if (part == FDTD3D_PART_INTERIOR) { kl0 = cut; kl1 = phase == 2 ? cut2 : hi; }
        else return fail(FDTD_ERR_ARG, "fused: bad part %d", part);
        const bool first_of_step = part == FDTD3D_PART_ALL || part == FDTD3D_PART_EDGE;
        if (first_of_step ? phase != 0 : (part == FDTD3D_PART_TOP ? phase != 1 : phase == 0))
            return fail(FDTD_ERR_STATE, "fused: a step is ALL, or EDGE [TOP] INTERIOR in this order");
        const bool mid_step = !first_of_step;
        K3<T> a = fused_args(n, mid_step);
        if (int rc = fused_go(a, kl0, kl1, part, s)) return rc;
        if (first_of_step) {
            for (int c = 0; c < 3; ++c) {
                std::swap(fld[c], alt[c]); std::swap(fld[6 + c], alt[6 + c]);
                for (int ax = 0; ax < 3; ++ax) for (int sd = 0; sd < 2; ++sd) std::swap(phE[c][ax][sd], phE2[c][ax][sd]);
            }
            ++flips;
        }
        phase = part == FDTD3D_PART_EDGE ? 1 : (part == FDTD3D_PART_TOP ? 2 : 0);
        return FDTD_OK;
    }

    // ---- two steps per launch (kernels3d_pair.hpp) ----------------------------------------------
    int wpbp = 16, chunk_pair = 0;       // 0: chosen per launch from the grid
    int pair_shape(int chunk_, int rows) override {
        if (chunk_ > 0) chunk_pair = chunk_;
        if (rows >= 6 && rows <= 16) wpbp = rows;
        return FDTD_OK;
    }
    // fdtd3d_step takes its steps two at a time when that is the faster way: a FUSED engine holding the whole grid,
    // and enough cells that workgroups of 16 waves x long chunks still fill the chip (FDTD_PAIR=0/1 overrides)
    bool pair_off = false;            // a two-step launch could not get its scratch memory: fdtd3d_step stays with single steps
    // whether two-step launches can run on this engine at all (whole grid or slab)
    bool pair_capable() override {
        if (!fused || pair_off) return false;
        for (int m = 0; m < 2; ++m) for (int c = 0; c < 3; ++c) for (int ax = 0; ax < 3; ++ax) for (int sd = 0; sd < 2; ++sd)
            if (phSize[m][c][ax][sd] >= ((size_t)1 << 31)) return false;       // the kernel indexes the CPML state with 32 bits
        return pair_scratch_ok();
    }
    bool pair_auto() override {
        if (d.z_lo != 0 || d.z_hi != d.Nz || !pair_capable()) return false;
        if (const char* e = getenv("FDTD_PAIR")) return atoi(e) != 0;
        bool pml = false;
        for (int ax = 0; ax < 3; ++ax) for (int sd = 0; sd < 2; ++sd) pml = pml || np[ax][sd] != 0;
        return (size_t)d.Nx * d.Ny * d.Nz >= (pml ? kPairMinCellsPml : kPairMinCells);
    }
    // measured, two steps per launch against one: without CPML 384^3 0.71 against 1.03 ms per step (200^3: 0.33 against
    // 0.24); with 20 layers on every face, where a third of the workgroups of a small grid run a layer loop, 384^3 1.39
    // against 1.31, 448^3 1.73 against 1.73, 512^3 2.03 against 2.23, 640^3 3.87 against 4.77
    static constexpr size_t kPairMinCells = (size_t)384 * 384 * 384;
    static constexpr size_t kPairMinCellsPml = (size_t)448 * 448 * 448;
    // x tiles of the two-step kernel: in a row, each overlapping the next by four columns; or, when both x faces carry
    // CPML layers and both strips fit one wave, tile 0 takes both ends of the row (its first `wrap_l` lanes the left
    // end, the others the columns from `wrap_r0` on, each part reaching four columns beyond its strip) and the other
    // tiles follow the left part: one x tile in nine instead of two then runs the x-strip loop (FDTD_PAIR_WRAP=0: never)
    struct PairTiles { int nbx, wrap_l, wrap_r0; };
    PairTiles pair_tiles() const {
        constexpr int PV = V / 2, TW = kLanes * PV, XS = TW - 4;
        const int cover = std::min(d.Nx + 1, px);            // columns that hold anything
        PairTiles t{cover > TW - 2 ? (cover - (TW - 2) + XS - 1) / XS + 1 : 1, 0, 0};
        const char* e = getenv("FDTD_PAIR_WRAP");
        if (t.nbx > 1 && np[0][0] > 0 && np[0][1] > 0 && !(e && atoi(e) == 0)) {
            const int r0 = xp0 - 4;                          // first column of the right part (xp0 is a multiple of 4)
            const int rend = std::min(px, round_up(cover + 2, 4));      // the right part ends in the padding, or at the wall
            const int nr = (rend - r0) / PV, nl = kLanes - nr;
            if (r0 >= 0 && nl * PV >= xwn + 4 && nl * PV <= r0) {
                const int from = nl * PV - 2, to = r0 + 2;   // what the tiles after tile 0 own
                t = PairTiles{1 + (to > from ? (to - from + XS - 1) / XS : 0), nl, r0};
            }
        }
        return t;
    }
    int pair_nbx() const { return pair_tiles().nbx; }
    // planes per workgroup: long chunks amortise the three extra iterations of a chunk, but the launch should
    // still have a few workgroups per CU
    // A CU holds one workgroup of the two-step kernel at a time, so a launch goes through the chip in rounds of as many
    // workgroups as there are CUs: of the chunk counts around the one above, take the one whose last round is fullest
    // (weighed against the three extra iterations every chunk costs).  512^3 without CPML: 5 chunks of 103 planes
    // (1000 workgroups, 3.9 rounds) run in 1.21 ms per step, 6 chunks of 86 (4.7 rounds) in 1.23.
    int pair_chunk() const {
        if (chunk_pair > 0) return chunk_pair;
        const int per_layer = pair_nbx() * ((ny1 + (wpbp - 3) - 1) / (wpbp - 3));
        const int want = (1024 + per_layer - 1) / per_layer;
        const int planes = std::max(1, nzp - 1);
        int cus = 0, dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        int best = std::max(8, std::min(128, (planes + want - 1) / want));
        double best_cost = 1e30;
        for (int nb = std::max(1, want - 1); nb <= want + 3; ++nb) {
            const int c = (planes + nb - 1) / nb;
            if (c > 128 || c < 8) continue;
            const double groups = (double)per_layer * ((planes + c - 1) / c);
            const double cost = std::ceil(groups / cus) * cus / groups * (c + 3.0) / c;
            if (cost < best_cost - 1e-9) { best_cost = cost; best = c; }
        }
        return best;
    }
    void pair_give_up() override { pair_off = true; (void)hipGetLastError(); }      // the failed allocation's error is not the next launch's
    void pair_info(int out[4]) override { out[0] = pair_auto() ? 1 : 0; out[1] = pair_chunk(); out[2] = wpbp; out[3] = V / 2; }
    // tile order of the two-step kernel: as dealt (0: x tiles fastest, round-robin over the XCDs) with the x tiles in a
    // row -- measured at 1024^3 with CPML: 12.3 ms against 12.7 with the x tiles of a row group kept on one XCD; with
    // both strips in tile 0 the row group per XCD is the better one (11.2-11.3 against 11.4-11.6, three runs each)
    int pair_remap(const PairTiles& t) const { return remap == kRemapAuto ? (t.wrap_l > 0 ? t.nbx : 0) : remap; }
    void pair_layout(int out[4]) override { const PairTiles t = pair_tiles(); out[0] = t.nbx; out[1] = t.wrap_l; out[2] = t.wrap_r0; out[3] = pair_remap(t); }
    int step_pair(int n, hipStream_t s) override { return step_pair_part(n, FDTD3D_PART_ALL, s); }
    // Cuts the planes [r0, r1) of a two-step launch into chunks; returns the number of workgroups (a.nbx, a.nby and
    // a.chunk must be set).
    unsigned pair_plan(K3<T>& a, int r0, int r1) const {
        a.kl0 = r0; a.kl1 = r1;
        // chunks that come near the z layers (within the four planes a chunk looks below itself, the one above) are cut
        // separately: everything else then runs the loop without z-layer code.  Local plane = global plane - zoff.
        a.zcut[0] = np[2][0] > 0 ? std::max(a.kl0, std::min(np[2][0] + 4 - zoff, a.kl1)) : a.kl0;
        a.zcut[1] = np[2][1] > 0 ? std::min(a.kl1, std::max(a.zcut[0], d.Nz - np[2][1] - 2 - zoff)) : a.kl1;
        const int len[3] = {a.zcut[0] - a.kl0, a.zcut[1] - a.zcut[0], a.kl1 - a.zcut[1]};
        int nb[3];
        for (int q = 0; q < 3; ++q) {
            a.zchunk[q] = (q == 1 || chunk_pair > 0) ? a.chunk : std::max(1, len[q]);      // a face segment is one chunk unless told otherwise
            nb[q] = (len[q] + a.zchunk[q] - 1) / a.zchunk[q];
        }
        a.znb[0] = nb[0]; a.znb[1] = nb[1];
        a.nbz = nb[0] + nb[1] + nb[2];
        return (unsigned)a.nbx * a.nby * a.nbz;
    }
    // whether a two-step launch over all owned planes stays within the scratch memory the kernel can address
    bool pair_scratch_ok() const {
        bool pml = false;
        for (int ax = 0; ax < 3; ++ax) for (int sd = 0; sd < 2; ++sd) pml = pml || np[ax][sd] != 0;
        if (!pml) return true;
        K3<T> a{};
        a.nbx = pair_tiles().nbx; a.nby = (ny1 + (wpbp - 3) - 1) / (wpbp - 3); a.chunk = pair_chunk();
        return fdtd3d_pair_scratch_fits((long long)pair_plan(a, 1, last_rank ? nzp : nzp - 1), wpbp, (int)sizeof(T) * 8) != 0;
    }
    // One launch of the two-step kernel over the owned planes [r0, r1).
    int pair_launch(K3<T> a, int r0, int r1, int part, bool pml, hipStream_t s) {
        if (r1 <= r0) return FDTD_OK;
        constexpr int PV = V / 2, TW = kLanes * PV;
        const unsigned nblk = pair_plan(a, r0, r1);
        const bool acc = a.acc != nullptr;         // the RMS window rides along: three planes of E1 and E2 rows in LDS instead of two
        const size_t lds = ((size_t)6 + (size_t)(acc ? 8 : 6) * wpbp * 2) * TW * sizeof(T);
        const size_t scr_need = pml ? (size_t)nblk * wpbp * 24 * TW : 0;
        // the kernel forms a wave's offset into the scratch memory in 32 bits
        if (!fdtd3d_pair_scratch_fits((long long)nblk, wpbp, (int)sizeof(T) * 8))
            return fail(FDTD_ERR_STATE, "two-step launch: %u workgroups need %zu elements of scratch memory, the kernel addresses 2^32; use longer chunks (fdtd3d_pair_shape)", nblk, scr_need);
        if (scr_need > pair_scr_elems) {
            // (launches of one stream run one after the other, so the new piece can replace the old one for all of them;
            // the old piece stays with the engine until it is destroyed)
            if (int rc = dalloc((void**)&pair_scr, scr_need * sizeof(T), 0)) return rc;
            pair_scr_elems = scr_need;
        }
        a.pair_scr = pair_scr;
        auto go = [&](auto kern) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, dim3(nblk), dim3(kLanes * wpbp), lds, s, a);
        };
        timer.begin(3, s, part);
        if (acc) {
            if (pml) { if (d.has_mu) go(k3_pair<T, true, true, PV, true>); else go(k3_pair<T, false, true, PV, true>); }
            else { if (d.has_mu) go(k3_pair<T, true, false, PV, true>); else go(k3_pair<T, false, false, PV, true>); }
        } else if (pml) { if (d.has_mu) go(k3_pair<T, true, true, PV>); else go(k3_pair<T, false, true, PV>); }
        else { if (d.has_mu) go(k3_pair<T, true, false, PV>); else go(k3_pair<T, false, false, PV>); }
        timer.end(s);
        FDTD_HIP(hipGetLastError());
        return FDTD_OK;
    }
    // Steps n and n+1 in one launch (PART_ALL), or in two parts so that a slab driver can send its edge planes while the
    // bulk is computed: PART_EDGE = the three lowest and the three highest owned planes (the buffers change roles here, as
    // with fdtd3d_fused), PART_INTERIOR = the rest.
    int pair_phase = 0;
    hipStream_t pair_stream = nullptr;
    T pair_acc_w = (T)0;              // step_observed: weight and flags of the window the next two-step launch advances
    int pair_acc_flags = -1;          // (-1: the launch leaves the window alone)
    int pair_acc_defer = -1;
    static constexpr int kPairEdge = 3;
    int step_pair_part(int n, int part, hipStream_t s) override {
        if (!fused) return fail(FDTD_ERR_STATE, "two-step launches need an engine created with FDTD3D_MODE_FUSED");
        bool pml = false;
        for (int ax = 0; ax < 3; ++ax) for (int sd = 0; sd < 2; ++sd) pml = pml || np[ax][sd] != 0;
        for (int m = 0; m < 2; ++m) for (int c = 0; c < 3; ++c) for (int ax = 0; ax < 3; ++ax) for (int sd = 0; sd < 2; ++sd)
            if (phSize[m][c][ax][sd] >= ((size_t)1 << 31)) return fail(FDTD_ERR_STATE, "two-step launches index the CPML state with 32 bits: an array of %zu elements is too large", phSize[m][c][ax][sd]);
        if (phase != 0) return fail(FDTD_ERR_STATE, "two-step launch in the middle of a split step");
        if (part != FDTD3D_PART_ALL && part != FDTD3D_PART_EDGE && part != FDTD3D_PART_INTERIOR) return fail(FDTD_ERR_ARG, "two-step launch: bad part %d", part);
        if ((part == FDTD3D_PART_INTERIOR) != (pair_phase == 1)) return fail(FDTD_ERR_STATE, "two-step launches: a pair of steps is ALL, or EDGE then INTERIOR");
        // all launches of an engine share one piece of scratch memory, indexed by workgroup: two of them must not run
        // at the same time, so both parts of a pair go on one stream
        if (part == FDTD3D_PART_INTERIOR && s != pair_stream) return fail(FDTD_ERR_STATE, "two-step launches: INTERIOR must be launched on the stream EDGE was launched on");
        K3<T> a = args();
        const bool s1 = n >= 0 && n < (int)wf.size(), s2 = n + 1 >= 0 && n + 1 < (int)wf.size();
        a.src_on = (s1 ? 1 : 0) | (s2 ? 2 : 0);
        a.jx = s1 ? a.dt * wf[n] : (T)0;
        a.jx2 = s2 ? a.dt * wf[n + 1] : (T)0;
        a.acc = nullptr; a.acc_snap = nullptr; a.acc_w = (T)0; a.acc_flags = 0; a.acc_defer_kl = -1;
        a.acc_defer_kl = -1;
        if (pair_acc_flags >= 0) { a.acc = obs_acc; a.acc_snap = obs_snap; a.acc_w = pair_acc_w; a.acc_flags = pair_acc_flags; a.acc_defer_kl = pair_acc_defer; }
        if (pml && !phM2_ready) {
            for (int c = 0; c < 3; ++c) for (int ax = 0; ax < 3; ++ax) for (int sd = 0; sd < 2; ++sd)
                if (phM[c][ax][sd])
                    if (int rc = dalloc((void**)&phM2[c][ax][sd], (phSize[1][c][ax][sd] + px) * sizeof(T))) return rc;
            phM2_ready = true;
        }
        for (int c = 0; c < 3; ++c) {
            a.Dn[c] = alt[c]; a.Bn[c] = alt[6 + c];
            for (int ax = 0; ax < 3; ++ax) for (int sd = 0; sd < 2; ++sd) { a.phEn[c][ax][sd] = phE2[c][ax][sd]; a.phMn[c][ax][sd] = phM2[c][ax][sd]; }
        }
        if (part == FDTD3D_PART_INTERIOR)       // the roles were exchanged when EDGE was launched
            for (int c = 0; c < 3; ++c) {
                std::swap(a.D[c], a.Dn[c]); std::swap(a.B[c], a.Bn[c]);
                for (int ax = 0; ax < 3; ++ax) for (int sd = 0; sd < 2; ++sd) { std::swap(a.phE[c][ax][sd], a.phEn[c][ax][sd]); std::swap(a.phM[c][ax][sd], a.phMn[c][ax][sd]); }
            }
        const PairTiles tiles = pair_tiles();
        a.nbx = tiles.nbx; a.wrap_l = tiles.wrap_l; a.wrap_r0 = tiles.wrap_r0;
        a.wpb = wpbp;
        a.nby = (ny1 + (wpbp - 3) - 1) / (wpbp - 3);
        a.chunk = pair_chunk();
        a.remap = pair_remap(tiles);
        const int lo = 1, hi = last_rank ? nzp : nzp - 1;         // the owned planes (on a slab: see fdtd3d_step_pairs)
        const int cut = std::min(lo + kPairEdge, hi), cut2 = std::max(cut, hi - kPairEdge);
        int rc = FDTD_OK;
        if (part == FDTD3D_PART_ALL) rc = pair_launch(a, lo, hi, part, pml, s);
        else if (part == FDTD3D_PART_EDGE) { rc = pair_launch(a, lo, cut, part, pml, s); if (!rc) rc = pair_launch(a, cut2, hi, part, pml, s); }
        else rc = pair_launch(a, cut, cut2, part, pml, s);
        if (rc) return rc;
        if (part != FDTD3D_PART_INTERIOR) {
            for (int c = 0; c < 3; ++c) {
                std::swap(fld[c], alt[c]); std::swap(fld[6 + c], alt[6 + c]);
                if (pml) for (int ax = 0; ax < 3; ++ax) for (int sd = 0; sd < 2; ++sd) { std::swap(phE[c][ax][sd], phE2[c][ax][sd]); std::swap(phM[c][ax][sd], phM2[c][ax][sd]); }
            }
            ++flips;
        }
        pair_phase = part == FDTD3D_PART_EDGE ? 1 : 0;
        pair_stream = s;
        return FDTD_OK;
    }

    int halo(int which, int comp, void** p, size_t* nbytes, int at_parity) override {
        if (comp < 0 || comp > 1) return fail(FDTD_ERR_ARG, "halo: comp must be 0 or 1");
        if (at_parity >= 0 && !fused && at_parity != 0) return fail(FDTD_ERR_ARG, "halo: only FUSED engines have a second buffer set");
        T* const* fld = (fused && at_parity >= 0 && at_parity != parity()) ? alt : this->fld;
        T* base = nullptr;
        int kl = 0;
        switch (which) {
            case FDTD3D_HALO_SEND_UP: base = fld[(full ? 9 : 6) + comp]; kl = nzl; break;       // H (or B), layer z_hi-1
            case FDTD3D_HALO_RECV_LO: base = fld[(full ? 9 : 6) + comp]; kl = 0; break;         // ghost z_lo-1
            case FDTD3D_HALO_SEND_DOWN: base = fld[(full ? 3 : 0) + comp]; kl = 1; break;       // E (or D), layer z_lo
            case FDTD3D_HALO_RECV_HI: base = fld[(full ? 3 : 0) + comp]; kl = nzl + 1; break;   // ghost z_hi
            default: return fail(FDTD_ERR_ARG, "halo: bad selector %d", which);
        }
        *p = base + (size_t)kl * plane;
        *nbytes = plane * sizeof(T);
        return FDTD_OK;
    }
};

int check_desc(const fdtd3d_desc* d) {
    if (!d) return fail(FDTD_ERR_ARG, "null descriptor");
    if (d->struct_size != (int32_t)sizeof(fdtd3d_desc)) return fail(FDTD_ERR_ARG, "descriptor size %d, library expects %zu", d->struct_size, sizeof(fdtd3d_desc));
    if (d->precision != 32 && d->precision != 64) return fail(FDTD_ERR_ARG, "precision must be 32 or 64");
    const int N[3] = {d->Nx, d->Ny, d->Nz};
    for (int a = 0; a < 3; ++a) {
        if (N[a] < 2) return fail(FDTD_ERR_ARG, "grid extent %d along axis %d is too small", N[a], a);
        const int nn = d->n_pml[2 * a], np = d->n_pml[2 * a + 1];
        if (nn < 0 || np < 0 || nn > kMaxLayers || np > kMaxLayers) return fail(FDTD_ERR_ARG, "CPML layer counts must be in [0,%d]", kMaxLayers);
        // the reference's low, middle and high sweeps only tile the axis without overlap under this condition
        if (nn + np >= N[a]) return fail(FDTD_ERR_ARG, "CPML layers (%d+%d) do not fit %d cells along axis %d", nn, np, N[a], a);
    }
    if (d->n_pml[0] > 0 && d->n_pml[1] > 0 && (d->Nx - d->n_pml[1]) / 4 * 4 < round_up(d->n_pml[0] + 1, 4))
        return fail(FDTD_ERR_ARG, "x extent %d too small for separate low/high CPML strips of %d and %d layers", d->Nx, d->n_pml[0], d->n_pml[1]);
    if (d->z_lo < 0 || d->z_hi > d->Nz || d->z_lo >= d->z_hi) return fail(FDTD_ERR_ARG, "bad slab [%d,%d) of %d layers", d->z_lo, d->z_hi, d->Nz);
    if (d->k_Jx < 0 || d->k_Jx > d->Nz) return fail(FDTD_ERR_ARG, "source plane %d outside [0,%d]", d->k_Jx, d->Nz);
    if (d->mode != FDTD3D_MODE_FULL && d->mode != FDTD3D_MODE_LEAN && d->mode != FDTD3D_MODE_FUSED) return fail(FDTD_ERR_ARG, "unknown storage mode %d", d->mode);
    if (!(d->dx > 0 && d->dy > 0 && d->dz > 0 && d->dt > 0 && d->mu0 > 0)) return fail(FDTD_ERR_ARG, "dx, dy, dz, dt, mu0 must be positive");
    return FDTD_OK;
}

}  // namespace

extern "C" {

int fdtd3d_create(const fdtd3d_desc* desc, fdtd3d_handle* out) {
    if (!out) return fail(FDTD_ERR_ARG, "null output handle");
    *out = nullptr;
    if (int rc = check_desc(desc)) return rc;
    int ndev = 0;
    FDTD_HIP(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(FDTD_ERR_HIP, "no HIP device visible");
    if (desc->device >= ndev) return fail(FDTD_ERR_ARG, "device %d requested, %d visible", desc->device, ndev);
    try {
        DeviceScope scope(desc->device);
        if (scope.status != hipSuccess) return fail(FDTD_ERR_HIP, "cannot select device %d: %s", desc->device, hipGetErrorString(scope.status));
        int dev = 0;
        FDTD_HIP(hipGetDevice(&dev));
        int rc;
        if (desc->precision == 32) {
            auto e = std::make_unique<Engine3<float>>();
            e->device = dev;
            if ((rc = e->init(*desc))) return rc;
            *out = e.release();
        } else {
            auto e = std::make_unique<Engine3<double>>();
            e->device = dev;
            if ((rc = e->init(*desc))) return rc;
            *out = e.release();
        }
    } catch (const std::bad_alloc&) {
        return fail(FDTD_ERR_HOST, "out of host memory");
    }
    return FDTD_OK;
}
int fdtd3d_destroy(fdtd3d_handle h) {
    if (!h) return FDTD_OK;
    DeviceScope scope(h->device);
    delete h;
    return FDTD_OK;
}
size_t fdtd3d_array_size(fdtd3d_handle h, int id) { return h ? h->array_size(id) : 0; }
int fdtd3d_set_array(fdtd3d_handle h, int id, const double* host, size_t n) { return entry(h, [&] { return h->set_array(id, host, n); }); }
int fdtd3d_get_array(fdtd3d_handle h, int id, double* host, size_t n) { return entry(h, [&] { return h->get_array(id, host, n); }); }
int fdtd3d_fill_materials(fdtd3d_handle h, uint64_t seed, double elo, double ehi, double mlo, double mhi) {
    return entry(h, [&] { return h->fill_materials(seed, elo, ehi, mlo, mhi); });
}
int fdtd3d_fill_reference_materials(fdtd3d_handle h) { return entry(h, [&] { return h->fill_reference_materials(); }); }
int fdtd3d_fill_array(fdtd3d_handle h, int id, uint64_t seed, double lo, double hi) { return entry(h, [&] { return h->fill_array(id, seed, lo, hi); }); }
int fdtd3d_step(fdtd3d_handle h, int n0, int count, void* stream) {
    return entry(h, [&]() -> int {
        if (h->d.z_lo != 0 || h->d.z_hi != h->d.Nz) return fail(FDTD_ERR_STATE, "fdtd3d_step needs the whole grid on this device; slab drivers use fdtd3d_half_e / fdtd3d_half_h or fdtd3d_fused");
        const bool pairs = h->pair_auto();
        const bool probed = h->probe_count() > 0;        // a probe is sampled after every single step: see pair_peek_probes
        for (int n = n0; n < n0 + count;) {
            if (pairs && n + 2 <= n0 + count && h->pair_auto()) {
                if (probed) if (int rc = h->pair_peek_probes(n, (hipStream_t)stream)) return rc;
                const int rc = h->step_pair(n, (hipStream_t)stream);
                if (rc == FDTD_OK) {
                    if (int rc2 = h->probe_record((hipStream_t)stream)) return rc2;
                    n += 2;
                    continue;
                }
                // device memory for the second B-side CPML buffers or the scratch ran out: the two-step kernel has not
                // been launched, the one-step kernel needs neither (and samples step n again)
                if (rc != FDTD_ERR_HIP) return rc;
                if (probed) h->probe_unrecord();
                h->pair_give_up();
            }
            if (int rc = h->step(n, (hipStream_t)stream)) return rc;
            if (int rc = h->probe_record((hipStream_t)stream)) return rc;
            ++n;
        }
        return FDTD_OK;
    });
}
int fdtd3d_step_pairs(fdtd3d_handle h, int n0, int pairs, void* stream) {
    return entry(h, [&]() -> int {
        const bool probed = h->probe_count() > 0;
        if (probed && (h->d.z_lo != 0 || h->d.z_hi != h->d.Nz))
            return fail(FDTD_ERR_STATE, "two-step launches on a slab cannot sample probes after the first of their steps");
        for (int q = 0; q < pairs; ++q) {
            if (probed) if (int rc = h->pair_peek_probes(n0 + 2 * q, (hipStream_t)stream)) return rc;
            if (int rc = h->step_pair(n0 + 2 * q, (hipStream_t)stream)) { if (probed) h->probe_unrecord(); return rc; }
            if (int rc = h->probe_record((hipStream_t)stream)) return rc;
        }
        return FDTD_OK;
    });
}
int fdtd3d_pair_part(fdtd3d_handle h, int n, int part, void* stream) {
    return entry(h, [&]() -> int {
        if (h->probe_count() > 0) return fail(FDTD_ERR_STATE, "two-step launches cannot sample probes after the first of their steps");
        return h->step_pair_part(n, part, (hipStream_t)stream);
    });
}
int fdtd3d_pair_info(fdtd3d_handle h, int out[4]) {
    if (!h || !out) return fail(FDTD_ERR_ARG, "null argument");
    h->pair_info(out);
    return FDTD_OK;
}
int fdtd3d_pair_capable(fdtd3d_handle h) { return (h && h->pair_capable()) ? 1 : 0; }
int fdtd3d_pair_layout(fdtd3d_handle h, int out[4]) {
    if (!h || !out) return fail(FDTD_ERR_ARG, "null argument");
    h->pair_layout(out);
    return FDTD_OK;
}
int fdtd3d_pair_shape(fdtd3d_handle h, int chunk, int rows) { return entry(h, [&] { return h->pair_shape(chunk, rows); }); }
int fdtd3d_fused(fdtd3d_handle h, int n, int part, void* stream) { return entry(h, [&] { return h->fused_part(n, part, (hipStream_t)stream); }); }
int fdtd3d_fused_top(fdtd3d_handle h, void* stream) { return entry(h, [&] { return h->fused_top((hipStream_t)stream); }); }
int fdtd3d_probe_add(fdtd3d_handle h, int id, int i, int j, int k) { return entry(h, [&] { return h->probe_add(id, i, j, k); }); }
int fdtd3d_probe_record(fdtd3d_handle h, void* stream) { return entry(h, [&] { return h->probe_record((hipStream_t)stream); }); }
int fdtd3d_probe_read(fdtd3d_handle h, double* out, size_t cap, int* n_samples, int* n_probes) { return entry(h, [&] { return h->probe_read(out, cap, n_samples, n_probes); }); }
int fdtd3d_probe_reset(fdtd3d_handle h, int drop_probes) { return entry(h, [&] { return h->probe_reset(drop_probes); }); }
int fdtd3d_spread_gaps(fdtd3d_handle h, double* gib, int cap) { return (h && (gib || cap == 0)) ? h->gaps(gib, cap) : 0; }
int fdtd3d_buffer_parity(fdtd3d_handle h) { return h ? h->parity() : 0; }
int fdtd3d_half_e(fdtd3d_handle h, int n, int part, void* stream) { return entry(h, [&] { return h->half_e(n, part, (hipStream_t)stream); }); }
int fdtd3d_half_h(fdtd3d_handle h, int part, void* stream) { return entry(h, [&] { return h->half_h(part, (hipStream_t)stream); }); }
int fdtd3d_halo(fdtd3d_handle h, int which, int comp, void** p, size_t* bytes) {
    if (!h || !p || !bytes) return fail(FDTD_ERR_ARG, "null argument");
    return h->halo(which, comp, p, bytes, -1);         // pointer arithmetic only: no device call
}
int fdtd3d_halo_at(fdtd3d_handle h, int parity, int which, int comp, void** p, size_t* bytes) {
    if (!h || !p || !bytes || parity < 0 || parity > 1) return fail(FDTD_ERR_ARG, "bad argument");
    return h->halo(which, comp, p, bytes, parity);
}
int fdtd3d_sumsq(fdtd3d_handle h, int id, double* out, void* stream) { return entry(h, [&] { return h->reduce(id, out, nullptr, (hipStream_t)stream); }); }
int fdtd3d_checksum(fdtd3d_handle h, int id, uint64_t* out, void* stream) { return entry(h, [&] { return h->reduce(id, nullptr, out, (hipStream_t)stream); }); }
int fdtd3d_reduce_range(fdtd3d_handle h, int id, int k_lo, int k_hi, double* sumsq, uint64_t* checksum, void* stream) {
    return entry(h, [&] { return h->reduce_range(id, k_lo, k_hi, sumsq, checksum, (hipStream_t)stream); });
}
int fdtd3d_planes(fdtd3d_handle h, int id, int k, int count, void** p, size_t* bytes) {
    if (!h || !p || !bytes) return fail(FDTD_ERR_ARG, "null argument");
    return h->planes(id, k, count, p, bytes);          // pointer arithmetic only: no device call
}
int fdtd3d_reduce_static(fdtd3d_handle h, const int* ids, int n_ids, double* sumsq, uint64_t* checksum, void* stream) {
    return entry(h, [&] { return h->reduce_static(ids, n_ids, sumsq, checksum, (hipStream_t)stream); });
}
size_t fdtd3d_device_bytes(fdtd3d_handle h) { return h ? h->device_bytes() : 0; }
int fdtd3d_grant_peer_access(fdtd3d_handle h, int peer_device) { return entry(h, [&] { return h->grant_access(peer_device); }); }
double fdtd3d_spread_gap_gib(fdtd3d_handle h) { return h ? h->spread_gap() : 0.0; }
int fdtd3d_launch_shape(fdtd3d_handle h, int out[6]) {
    if (!h || !out) return fail(FDTD_ERR_ARG, "null argument");
    h->launch_shape(out);
    return FDTD_OK;
}
int fdtd3d_tune(fdtd3d_handle h, int chunk, int remap, int rows) { return entry(h, [&] { return h->tune(chunk, remap, rows); }); }
int fdtd3d_observe_accumulate(fdtd3d_handle h, double weight, int reset, void* stream) { return entry(h, [&] { return h->observe_accumulate(weight, reset, (hipStream_t)stream); }); }
int fdtd3d_observe_rms(fdtd3d_handle h, double* host, size_t n) { return entry(h, [&] { return h->observe_rms(host, n); }); }
int fdtd3d_observe_set_range(fdtd3d_handle h, int k_lo, int k_hi) { return entry(h, [&] { return h->observe_set_range(k_lo, k_hi); }); }
int fdtd3d_pair_part_observed(fdtd3d_handle h, int n, int part, double weight, int window, int defer_top, void* stream) {
    return entry(h, [&] { return h->pair_part_observed(n, part, weight, window, defer_top, (hipStream_t)stream); });
}
int fdtd3d_observe_finish_pair(fdtd3d_handle h, int n, double weight, int window, int defer_top, void* stream) {
    return entry(h, [&] { return h->observe_finish_pair(n, weight, window, defer_top, (hipStream_t)stream); });
}
int fdtd3d_observe_keep_window(fdtd3d_handle h) { return entry(h, [&] { return h->observe_keep_window(); }); }
int fdtd3d_observe_rms_window(fdtd3d_handle h, double* host, size_t n) { return entry(h, [&] { return h->observe_rms_window(host, n); }); }
int fdtd3d_step_observed(fdtd3d_handle h, int n0, int count, double weight, int window, void* stream) {
    return entry(h, [&]() -> int {
        if (h->d.z_lo != 0 || h->d.z_hi != h->d.Nz) return fail(FDTD_ERR_STATE, "fdtd3d_step_observed needs the whole grid on this device; slab drivers step and call fdtd3d_observe_accumulate themselves");
        return h->step_observed(n0, count, weight, window, (hipStream_t)stream);
    });
}
int fdtd3d_observe_frame(fdtd3d_handle h, double* ex_center, double* e_mag, size_t n, void* stream) { return entry(h, [&] { return h->observe_frame(ex_center, e_mag, n, (hipStream_t)stream); }); }
int fdtd3d_kernel_timing(fdtd3d_handle h, int enable) {
    if (!h) return fail(FDTD_ERR_ARG, "null handle");
    h->timer.enabled = enable != 0;
    return FDTD_OK;
}
int fdtd3d_kernel_time(fdtd3d_handle h, int kind, int* launches, double* total_ms) {
    return entry(h, [&] { return h->timer.collect(kind, launches, total_ms); });
}
int fdtd3d_kernel_timeline(fdtd3d_handle h, int cap, int* kinds, int* parts, double* start_ms, double* dur_ms, int* n_launches) {
    return entry(h, [&] { return h->timer.timeline(cap, kinds, parts, start_ms, dur_ms, n_launches); });
}

}  // extern "C"
