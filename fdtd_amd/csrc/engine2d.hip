// engine2d.hip -- host side of the 2-D engine: device state, array exchange, launches, C ABI.
#include "kernels2d_fused.hpp"
#include "kernels2d_pair.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <utility>

using namespace fdtd;

namespace fdtd {
template <typename T>
__global__ void k2_sumsq(const T* arr, int Nx, int px, int rows, double* partial) {
    const size_t total = (size_t)rows * px;
    double s = 0;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        if ((int)(t % px) >= Nx) continue;
        const T v = arr[t];
        s += (double)v * (double)v;
    }
    __shared__ double sh[256];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}
// flags[k][tile] = 1 if any of C1, C2, C3 is non-zero (or NaN) in the 64-lane tile of row k
template <typename T, int V>
__global__ void k2_flags(const T* c1, const T* c2, const T* c3, unsigned char* flags, int px, int rows, int ntx) {
    const int lane = threadIdx.x & 63;
    const size_t tile = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (tile >= (size_t)rows * ntx) return;
    const int k = (int)(tile / ntx), tx = (int)(tile % ntx);
    const size_t idx = (size_t)k * px + ((size_t)tx * 64 + lane) * V;
    bool any = false;
    for (int e = 0; e < V; ++e) any = any || !(c1[idx + e] == (T)0) || !(c2[idx + e] == (T)0) || !(c3[idx + e] == (T)0);
    const unsigned long long m = __ballot(any);
    if (lane == 0) flags[tile] = m != 0ull;
}

// acc(k,i) (+)= Ey(i,k)^2 * weight, device layout [k][px]
template <typename T>
__global__ void k2_observe(const T* ey, T* acc, int Nx, int Nz, int px, T weight, int reset) {
    const size_t total = (size_t)Nz * px;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        if ((int)(t % px) >= Nx) continue;
        const T e = ey[t];
        const T old = reset ? (T)0 : acc[t];
        acc[t] = old + e * e * weight;
    }
}

// Every fp32 bit pattern of the class of `d` through the short division and through the IEEE one; counts disagreements.
__global__ void k_const_div_verify(ConstDiv<float> d, unsigned long long* bad) {
    unsigned long long mine = 0;
    const unsigned long long total = 1ull << 32;
    for (unsigned long long u = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((unsigned)u);
        if (!const_div_in_class(x, d)) continue;
        const float q = const_div_short(x, d), want = x / d.c;
        if (__float_as_uint(q) != __float_as_uint(want)) ++mine;
    }
    if (mine) atomicAdd(bad, mine);
}
}  // namespace fdtd

namespace {
// The constant's reciprocal and class bounds, and -- fp32 only -- whether the short division has been proven equal to the
// IEEE one for every numerator of the class on the current device (cached per constant; FDTD_CONST_DIV=0 turns it off).
template <typename T> int make_const_div(double c, fdtd::ConstDiv<T>& out);
template <> int make_const_div<double>(double c, fdtd::ConstDiv<double>& out) { out = {c, 1.0 / c, 0.0, 0.0, 0}; return FDTD_OK; }
template <> int make_const_div<float>(double cd, fdtd::ConstDiv<float>& out) {
    const float c = (float)cd;
    out = {c, 1.0f / c, 0.0f, 0.0f, 0};
    static std::vector<std::pair<unsigned, int>> cache;       // (bits of c, proven)
    static std::mutex cache_lock;
    std::lock_guard<std::mutex> guard(cache_lock);
    const char* env = getenv("FDTD_CONST_DIV");
    if ((env && atoi(env) == 0) || !(c > 0.0f) || !std::isfinite(c) || !std::isfinite(out.r)) return FDTD_OK;
    // the class: numerator, quotient and remainder well inside the normal range
    out.lo = 0x1p-80f * std::max(1.0f, c);
    out.hi = 0x1p+80f * std::min(1.0f, c);
    if (!(out.lo < out.hi)) return FDTD_OK;
    unsigned bits;
    std::memcpy(&bits, &c, 4);
    for (const auto& e : cache) if (e.first == bits) { out.fast = e.second; return FDTD_OK; }
    unsigned long long* bad_d = nullptr;
    unsigned long long bad = 1;
    FDTD_HIP(hipMalloc((void**)&bad_d, sizeof bad));
    FDTD_HIP(hipMemset(bad_d, 0, sizeof bad));
    out.fast = 1;                                               // (the verifier evaluates the class with the bounds above)
    hipLaunchKernelGGL(fdtd::k_const_div_verify, dim3(256 * 64), dim3(256), 0, 0, out, bad_d);
    FDTD_HIP(hipGetLastError());
    FDTD_HIP(hipMemcpy(&bad, bad_d, sizeof bad, hipMemcpyDeviceToHost));
    FDTD_HIP(hipFree(bad_d));
    out.fast = bad == 0 ? 1 : 0;
    cache.emplace_back(bits, out.fast);
    return FDTD_OK;
}
}  // namespace

struct fdtd2d_engine {
    virtual ~fdtd2d_engine() {}
    virtual size_t array_size(int id) = 0;
    virtual int set_array(int id, const double* host, size_t n) = 0;
    virtual int get_array(int id, double* host, size_t n) = 0;
    virtual int step(int n0, int count, hipStream_t s) = 0;
    virtual int sumsq(int id, double* out, hipStream_t s) = 0;
    virtual size_t device_bytes() = 0;
    virtual int observe_accumulate(double weight, int reset, hipStream_t s) = 0;
    virtual int observe_rms(double* host, size_t n) = 0;
    virtual int step_observed(int n0, int count, double weight, int window, hipStream_t s) = 0;
    virtual int observe_rms_window(double* host, size_t n) = 0;
    virtual int probe_add(int id, int i, int k) = 0;
    virtual int probe_record(hipStream_t s) = 0;
    virtual int probe_read(double* out, size_t cap, int* n_samples, int* n_probes) = 0;
    virtual int probe_reset(int drop_probes) = 0;
    fdtd2d_desc d;
    KernelTimer timer;
    int device = -1;                 // HIP device the state lives on
};

namespace {

template <typename T>
struct Engine2 : fdtd2d_engine {
    static constexpr int V = VecWidth<T>::V;
    static constexpr int PV = V / 2;          // elements per lane in the two-step kernel (8 bytes)
    int px = 0, nz1 = 0, np[2][2], xwn = 0, xwp = 0, xp0 = 0;
    size_t fsize = 0, bytes = 0;
    T* Dy = nullptr; T* Bx = nullptr; T* Bz = nullptr;
    T* Fa[3]; T* Fb[3]; T* Sa[3]; T* Sb[3];     // a = holds sample n-1 ("current"), b = sample n-2
    T* C[3];
    T* ph_z[2][2];      // [0 = ey, 1 = mx][side]   layout [l][px]
    T* ph_x[2][2];      // [0 = ey, 1 = mz][side]   layout [k][strip]
    T* tab_d = nullptr; T* xrow_d = nullptr; double* red_d = nullptr;
    unsigned char* disp_d = nullptr;
    // fused stepping: second Dy/Bx/Bz and D-side CPML buffers, a third Ey/SEy sample buffer, flags of
    // the overlapping row segments, a scratch field for downloads of fields that only exist in registers
    bool fused = false;
    T* Dy2 = nullptr; T* Bx2 = nullptr; T* Bz2 = nullptr;
    T* Ef = nullptr; T* SEf = nullptr;
    T* ph_z2[2] = {nullptr, nullptr}; T* ph_x2[2] = {nullptr, nullptr};
    unsigned char* disp2_d = nullptr;
    T* tmp_field = nullptr;
    T* xrow2_d = nullptr;
    int ntx2 = 0, chunk_f = 2;        // rows per wave in the fused kernel (2 is the optimum at 8192^2 with four stacked waves per workgroup)
    // two steps per launch (k2_pair): flags of its row segments, the B-side CPML state's second buffers, and a second
    // pair of sample and sum buffers per field (all allocated on the first two-step launch)
    int ntx3 = 0, chunk_p = 8, pair_env = -1;      // rows per wave: 8 is the optimum at 8192^2 (4-6: 0.30-0.32 ms per step, 8: 0.277, 12: 0.284, 19: 0.31, 38: 0.35)
    unsigned char* disp3_d = nullptr;
    bool any_disp = false, pair_ready = false, pair_allocated = false;
    T* pm_z2[2] = {nullptr, nullptr}; T* pm_x2[2] = {nullptr, nullptr};
    T* Fx[3] = {nullptr, nullptr, nullptr}; T* Fy[3] = {nullptr, nullptr, nullptr};
    T* Sx[3] = {nullptr, nullptr, nullptr}; T* Sy[3] = {nullptr, nullptr, nullptr};
    std::vector<double> tab_h;
    std::vector<T> Jy;
    bool remap_set = false;        // FDTD_XCD_REMAP given: it overrides the per-kernel default tile order
    int chunk = 1, remap = 1;      // rows per wave: at 8192^2 one row per wave is fastest (the row below comes from L2, and more waves are in flight)
    std::vector<void*> allocs;

    ~Engine2() override {
        if (lz_stream) { (void)hipStreamSynchronize(lz_stream); (void)hipStreamDestroy(lz_stream); (void)hipEventDestroy(lz_fork); (void)hipEventDestroy(lz_join); }
        for (void* p : allocs) (void)hipFree(p);
        probes.release(); timer.release();
    }

    // ---- probe time series: the sample arrays change places every step, so a probe names a slot of
    // this table (0-2 Dy,Bx,Bz; 3-5 current sample; 6-8 older sample; 9-11 / 12-14 the same for S; 15-17 C)
    ProbeLog<T> probes;
    int probe_add(int id, int i, int k) override {
        Map m;
        if (!((id >= 0 && id <= 20) || (id >= FDTD2D_C1 && id <= FDTD2D_C3)) || !make_map(id, m)) return fail(FDTD_ERR_ARG, "probe_add: %d is not a field array", id);
        if (i < 0 || i >= m.n0 || k < 0 || k >= m.n1) return fail(FDTD_ERR_ARG, "probe_add: cell (%d,%d) outside array %d", i, k, id);
        const long long off = dev_off(m, i, k);
        if (off < 0) return fail(FDTD_ERR_ARG, "probe_add: column %d of array %d is the wall column, always zero", i, id);
        int slot;
        if (id >= FDTD2D_C1) slot = 15 + (id - FDTD2D_C1);
        else { const int f = id / 7, r = id % 7; slot = r == 0 ? f : (r <= 2 ? 3 + f : (r == 3 ? 6 + f : (r <= 5 ? 9 + f : 12 + f))); }
        ProbeSpec sp{slot, -1, off, 1.0};
        if (fused && id <= 20 && (id % 7 == 1 || id % 7 == 2)) {
            // the current Ey / Hx / Hz of a cell without Lorentz coefficients only exists as Dy/eps0 (B/mu0)
            const int f = id / 7;
            T c[3];
            for (int q = 0; q < 3; ++q) FDTD_HIP(hipMemcpy(&c[q], C[q] + off, sizeof(T), hipMemcpyDeviceToHost));
            if (c[0] == (T)0 && c[1] == (T)0 && c[2] == (T)0 && k < d.Nz) { sp.num = f; sp.den = -2; sp.sdiv = f == 0 ? d.eps0 : d.mu0; }
        }
        return probes.add(sp);
    }
    int probe_record(hipStream_t s) override {
        if (probes.specs.empty()) return FDTD_OK;
        T* row = nullptr;
        if (int rc = probes.next_row(&row, (int)Jy.size() + 8)) return rc;
        PtrTable<T, 18> t;
        t.p[0] = Dy; t.p[1] = Bx; t.p[2] = Bz;
        for (int f = 0; f < 3; ++f) { t.p[3 + f] = Fa[f]; t.p[6 + f] = Fb[f]; t.p[9 + f] = Sa[f]; t.p[12 + f] = Sb[f]; t.p[15 + f] = C[f]; }
        hipLaunchKernelGGL((k_probe<T, 18>), dim3(1), dim3(kMaxProbes), 0, s, t, probes.specs_d, (int)probes.specs.size(), row);
        FDTD_HIP(hipGetLastError());
        return FDTD_OK;
    }
    int probe_read(double* out, size_t cap, int* n_samples, int* n_probes) override { return probes.read(out, cap, n_samples, n_probes); }
    int probe_reset(int drop_probes) override { probes.reset(drop_probes != 0); return FDTD_OK; }

    // Arrays that are streamed at the same time stream faster when they lie far apart in physical memory (measured
    // at 8192^2, fused stepping: 0.345 against 0.363 ms per step with 10 GiB between them; 2 or 6 GiB change nothing).
    // While a large engine is set up, every field-sized allocation is therefore preceded by a throw-away one of
    // `gap_bytes`, all of which are freed again at the end of init(): the driver hands memory out in order.
    // FDTD2D_GAP_MB sets the gap (0: none).
    std::vector<void*> gap_allocs;
    size_t gap_bytes = 0;
    int dalloc(void** p, size_t nbytes) {
        if (nbytes == 0) { *p = nullptr; return FDTD_OK; }
        if (gap_bytes && nbytes >= ((size_t)64 << 20)) {
            void* g = nullptr;
            if (hipMalloc(&g, gap_bytes) == hipSuccess) gap_allocs.push_back(g); else (void)hipGetLastError();     // no room: no gap here
        }
        FDTD_HIP(hipMalloc(p, nbytes));
        allocs.push_back(*p);
        bytes += nbytes;
        FDTD_HIP(hipMemset(*p, 0, nbytes));
        return FDTD_OK;
    }
    int init(const fdtd2d_desc& dd) {
        d = dd;
        for (int a = 0; a < 2; ++a) for (int s = 0; s < 2; ++s) np[a][s] = d.n_pml[a * 2 + s];
        px = round_up(d.Nx, kRowAlign);
        nz1 = d.Nz + 1;
        fsize = (size_t)nz1 * px;
        xwn = np[0][0] > 0 ? round_up(np[0][0] + 1, 4) : 0;
        xp0 = (d.Nx - np[0][1]) / 4 * 4;
        xwp = np[0][1] > 0 ? round_up(d.Nx - xp0, 4) : 0;
        if (const char* e = getenv("FDTD_Z_CHUNK2D")) { int c = atoi(e); if (c >= 1 && c <= 65536) chunk = c; }
        if (const char* e = getenv("FDTD_XCD_REMAP")) { int v = atoi(e); if (v >= 0 && v < 65536) { remap = v; remap_set = true; } }
        int rc;
        const size_t fb = (fsize + px) * sizeof(T);
        struct GapGuard { std::vector<void*>& v; ~GapGuard() { for (void* g : v) (void)hipFree(g); v.clear(); } } gap_guard{gap_allocs};
        if (fb >= ((size_t)64 << 20)) {
            const size_t n_big = 18 + (d.mode == FDTD2D_MODE_FUSED ? 6 : 0), reserve = (size_t)4 << 30;
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
            const size_t room = free_b > n_big * fb + reserve ? free_b - n_big * fb - reserve : 0;
            gap_bytes = std::min((size_t)10 << 30, room / n_big);
            if (const char* e = getenv("FDTD2D_GAP_MB")) gap_bytes = std::min((size_t)std::max(0ll, atoll(e)) << 20, room / n_big);
            if (gap_bytes < ((size_t)1 << 30)) gap_bytes = 0;
        }
        if ((rc = dalloc((void**)&Dy, fb)) || (rc = dalloc((void**)&Bx, fb)) || (rc = dalloc((void**)&Bz, fb))) return rc;
        for (int f = 0; f < 3; ++f) {
            if ((rc = dalloc((void**)&Fa[f], fb)) || (rc = dalloc((void**)&Fb[f], fb)) || (rc = dalloc((void**)&Sa[f], fb)) ||
                (rc = dalloc((void**)&Sb[f], fb)) || (rc = dalloc((void**)&C[f], fb))) return rc;
        }
        for (int w = 0; w < 2; ++w) for (int s = 0; s < 2; ++s) {
            ph_z[w][s] = ph_x[w][s] = nullptr;
            if (np[1][s] && (rc = dalloc((void**)&ph_z[w][s], ((size_t)np[1][s] * px + px) * sizeof(T)))) return rc;
            if (np[0][s] && (rc = dalloc((void**)&ph_x[w][s], ((size_t)nz1 * (s ? xwp : xwn) + px) * sizeof(T)))) return rc;
        }
        tab_h.assign((size_t)2 * 2 * 6 * kMaxLayers, 0.0);
        if ((rc = dalloc((void**)&tab_d, tab_h.size() * sizeof(T)))) return rc;
        if ((rc = dalloc((void**)&xrow_d, (size_t)6 * px * sizeof(T)))) return rc;
        if ((rc = dalloc((void**)&red_d, 1024 * sizeof(double)))) return rc;
        if ((rc = dalloc((void**)&disp_d, (size_t)nz1 * (px / (kLanes * V)) + 64))) return rc;
        fused = d.mode == FDTD2D_MODE_FUSED;
        if (fused) {
            if (const char* e = getenv("FDTD_Z_CHUNK2D")) { int c = atoi(e); if (c >= 1 && c <= 65536) chunk_f = c; }
            ntx2 = (d.Nx + (kLanes - 1) * V - 1) / ((kLanes - 1) * V);
            if ((rc = dalloc((void**)&Dy2, fb)) || (rc = dalloc((void**)&Bx2, fb)) || (rc = dalloc((void**)&Bz2, fb)) ||
                (rc = dalloc((void**)&Ef, fb)) || (rc = dalloc((void**)&SEf, fb)) || (rc = dalloc((void**)&tmp_field, fb))) return rc;
            for (int s = 0; s < 2; ++s) {
                if (np[1][s] && (rc = dalloc((void**)&ph_z2[s], ((size_t)np[1][s] * px + px) * sizeof(T)))) return rc;
                if (np[0][s] && (rc = dalloc((void**)&ph_x2[s], ((size_t)nz1 * (s ? xwp : xwn) + px) * sizeof(T)))) return rc;
            }
            if ((rc = dalloc((void**)&disp2_d, (size_t)nz1 * ntx2 + 64))) return rc;
            // x tiles of the two-step kernel (8 bytes per lane): tile t owns the lanes [HL, 64 - HL) of its 64 (tile 0 its first ones as well)
            const int packs = (d.Nx + PV - 1) / PV, own_lanes = kLanes - 2 * pair2d_halo_lanes<PV>(), reach = kLanes - pair2d_halo_lanes<PV>();
            ntx3 = packs <= reach ? 1 : (packs - reach + own_lanes - 1) / own_lanes + 1;
            if ((rc = dalloc((void**)&disp3_d, (size_t)nz1 * ntx3 + 64))) return rc;
            if (const char* e = getenv("FDTD_Z_CHUNK2D_PAIR")) { int c = atoi(e); if (c >= 1 && c <= 65536) chunk_p = c; }
            if (const char* e = getenv("FDTD2D_PAIR")) pair_env = atoi(e) != 0 ? 1 : 0;
            if ((rc = make_const_div<T>(d.mu0, cd_mu)) || (rc = make_const_div<T>(d.eps0, cd_eps))) return rc;
            if ((rc = dalloc((void**)&xrow2_d, (size_t)6 * px * sizeof(T)))) return rc;
            if ((rc = upload_tables())) return rc;
        }
        Jy.assign(d.n_wf > 0 ? d.n_wf : 0, (T)0);
        gap_bytes = 0;                       // later allocations (observables) are plain ones
        return FDTD_OK;
    }
    size_t device_bytes() override { return bytes; }

    // ---- fused mode: fields that only exist in registers during a step, formed on request
    // family 0 = Ey (from Dy/eps0), 1 = Hx (Bx/mu0), 2 = Hz (Bz/mu0); rows the step never touches keep the stored sample
    const T* current_field(int f, hipStream_t s) {
        if (!fused) return Fa[f];
        const T* raw = f == 0 ? Dy : (f == 1 ? Bx : Bz);
        hipLaunchKernelGGL(k2_materialize<T>, dim3(2048), dim3(256), 0, s, raw, (const T*)Fa[f], (const T*)C[0], (const T*)C[1], (const T*)C[2], tmp_field,
                           (T)(f == 0 ? d.eps0 : d.mu0), px, nz1, d.Nz);
        return tmp_field;
    }
    T* obs_acc = nullptr;
    int observe_accumulate(double weight, int reset, hipStream_t s) override {
        int rc;
        if (!obs_acc && (rc = dalloc((void**)&obs_acc, fsize * sizeof(T)))) return rc;
        hipLaunchKernelGGL(k2_observe<T>, dim3(2048), dim3(256), 0, s, current_field(0, s), obs_acc, d.Nx, d.Nz, px, (T)weight, reset);
        FDTD_HIP(hipGetLastError());
        return FDTD_OK;
    }
    int download_window(const T* src, double* host) {
        FDTD_HIP(hipDeviceSynchronize());
        std::vector<T> st(fsize);
        FDTD_HIP(hipMemcpy(st.data(), src, st.size() * sizeof(T), hipMemcpyDeviceToHost));
        for (int i = 0; i < d.Nx; ++i) for (int k = 0; k < d.Nz; ++k) host[(size_t)i * d.Nz + k] = std::sqrt((double)st[(size_t)k * px + i]);
        return FDTD_OK;
    }
    int observe_rms(double* host, size_t n) override {
        if (n != (size_t)d.Nx * d.Nz || !host) return fail(FDTD_ERR_ARG, "observe_rms: expected %zu values", (size_t)d.Nx * d.Nz);
        if (!obs_acc) return fail(FDTD_ERR_STATE, "observe_rms: nothing accumulated yet");
        return download_window(obs_acc, host);
    }
    // ---- the loop body of FDTD2D::run (FDTD2D.cpp:510-522) for a run of steps: step, advance the window, keep a
    // window that completes.  Steps that go two per launch advance the window inside the launch.
    T* obs_snap = nullptr;
    bool obs_have_window = false;
    T pair_acc_w = (T)0;
    int pair_acc_flags = -1;          // (-1: the next two-step launch leaves the window alone)
    int step_observed(int n0, int count, double weight, int window, hipStream_t s) override {
        if (window < 1) return fail(FDTD_ERR_ARG, "step_observed: a window must have at least one step");
        if (count < 0) return fail(FDTD_ERR_ARG, "step_observed: negative step count");
        int rc;
        if (!obs_acc && (rc = dalloc((void**)&obs_acc, fsize * sizeof(T)))) return rc;
        if (!obs_snap && (rc = dalloc((void**)&obs_snap, fsize * sizeof(T)))) return rc;
        const bool pairs = pair_wanted();
        for (int n = n0; n < n0 + count;) {
            const bool two = pairs && n + 1 < n0 + count;
            const int last = two ? n + 1 : n;
            if (two) {
                const bool done1 = (n + 1) % window == 0;
                pair_acc_w = (T)weight;
                pair_acc_flags = (n % window == 0 ? 1 : 0) | ((n + 1) % window == 0 ? 2 : 0) | (done1 ? 4 : 0);
                rc = step_pair(n, s);
                pair_acc_flags = -1;
                if (rc) return rc;
                if (done1) obs_have_window = true;
            } else {
                if ((rc = step(n, 1, s)) || (rc = observe_accumulate(weight, n % window == 0, s))) return rc;
            }
            if ((last + 1) % window == 0) {          // the window is complete as it stands
                FDTD_HIP(hipMemcpyAsync(obs_snap, obs_acc, fsize * sizeof(T), hipMemcpyDeviceToDevice, s));
                obs_have_window = true;
            }
            n = last + 1;
        }
        return FDTD_OK;
    }
    int observe_rms_window(double* host, size_t n) override {
        if (n != (size_t)d.Nx * d.Nz || !host) return fail(FDTD_ERR_ARG, "observe_rms_window: expected %zu values", (size_t)d.Nx * d.Nz);
        if (!obs_have_window) return fail(FDTD_ERR_STATE, "observe_rms_window: no window has been completed yet");
        return download_window(obs_snap, host);
    }

    struct Map { int n0 = 0, n1 = 0; T* dev = nullptr; T* dev2 = nullptr; size_t dev_elems = 0; int kind = -1, s = 0; bool mag = false; int fam = -1, slot = -1; };
    bool make_map(int id, Map& m) {
        if (id >= 0 && id <= 20) {
            const int f = id / 7, r = id % 7;
            m.n0 = d.Nx + (f != 2); m.n1 = d.Nz + (f != 1);
            m.kind = 0; m.dev_elems = fsize;
            m.fam = f; m.slot = r;
            if (r == 0) { m.dev = f == 0 ? Dy : (f == 1 ? Bx : Bz); m.dev2 = f == 0 ? Dy2 : (f == 1 ? Bx2 : Bz2); }
            else if (r == 1 || r == 2) { m.dev = Fa[f]; if (f == 0) m.dev2 = Ef; }
            else if (r == 3) m.dev = Fb[f];
            else if (r == 4 || r == 5) m.dev = Sa[f];
            else m.dev = Sb[f];
            return true;
        }
        if (id >= FDTD2D_C1 && id <= FDTD2D_C3) { m.n0 = d.Nx; m.n1 = d.Nz; m.kind = 0; m.dev = C[id - FDTD2D_C1]; m.dev_elems = fsize; return true; }
        if (id >= FDTD2D_PHI && id < FDTD2D_PHI + 8) {
            const int w = (id - FDTD2D_PHI) / 2; m.s = (id - FDTD2D_PHI) % 2;
            m.mag = w >= 2;
            if (w == 0 || w == 2) { m.kind = 1; m.n0 = d.Nx + 1; m.n1 = np[1][m.s]; m.dev = ph_z[w / 2][m.s]; m.dev_elems = (size_t)np[1][m.s] * px; if (w == 0) m.dev2 = ph_z2[m.s]; }
            else { m.kind = 2; m.n0 = np[0][m.s]; m.n1 = d.Nz + 1; m.dev = ph_x[w / 2][m.s]; m.dev_elems = (size_t)nz1 * (m.s ? xwp : xwn); if (w == 1) m.dev2 = ph_x2[m.s]; }
            return true;
        }
        return false;
    }
    long long dev_off(const Map& m, int p0, int p1) const {
        switch (m.kind) {
            case 0: return p0 >= d.Nx ? -1 : (long long)p1 * px + p0;       // canonical (i,k) -> [k][i]
            case 1: return p0 >= d.Nx ? -1 : (long long)p1 * px + p0;       // canonical (i,l) -> [l][i]
            case 2: {                                                        // canonical (l,k) -> [k][column - origin]
                const int col = m.s ? d.Nx - np[0][1] + p0 : (m.mag ? p0 : 1 + p0);
                if (col < 0 || col >= d.Nx) return -1;
                return m.s ? (long long)p1 * xwp + (col - xp0) : (long long)p1 * xwn + col;
            }
        }
        return -1;
    }
    size_t array_size(int id) override {
        Map m;
        if (make_map(id, m)) return (size_t)m.n0 * m.n1;
        if (id >= FDTD2D_TAB && id < FDTD2D_TAB + 24) return np[(id - FDTD2D_TAB) / 12][((id - FDTD2D_TAB) / 6) % 2];
        if (id == FDTD2D_JY) return Jy.size();
        return 0;
    }
    int upload_tables() {
        std::vector<T> t(tab_h.size());
        for (size_t q = 0; q < t.size(); ++q) t[q] = (T)tab_h[q];
        FDTD_HIP(hipMemcpy(tab_d, t.data(), t.size() * sizeof(T), hipMemcpyHostToDevice));
        std::vector<T> xr((size_t)6 * px, (T)0);
        auto tb = [&](int s, int kind, int l) { return (T)tab_h[((size_t)s * 6 + kind) * kMaxLayers + l]; };
        for (int i = 0; i < d.Nx; ++i) {
            int l;
            int z = zone_e(i, d.Nx, np[0][0], np[0][1], l);
            if (z == Z_NEG || z == Z_POS) for (int q = 0; q < 3; ++q) xr[(size_t)q * px + i] = tb(z == Z_POS, q, l);
            z = zone_h(i, d.Nx, np[0][0], np[0][1], false, l);
            if (z == Z_NEG || z == Z_POS) for (int q = 3; q < 6; ++q) xr[(size_t)q * px + i] = tb(z == Z_POS, q, l);
        }
        FDTD_HIP(hipMemcpy(xrow_d, xr.data(), xr.size() * sizeof(T), hipMemcpyHostToDevice));
        if (fused) {
            // the fused kernel's x terms are branch-free: plain columns carry (0, 0, dt/dx), columns the
            // reference's sweeps skip (0, 0, 0)
            const T cmid = (T)(d.dt / d.dx);
            for (int i = 0; i < d.Nx; ++i) {
                int l;
                if (zone_e(i, d.Nx, np[0][0], np[0][1], l) == Z_MID) xr[(size_t)2 * px + i] = cmid;
                if (zone_h(i, d.Nx, np[0][0], np[0][1], false, l) == Z_MID) xr[(size_t)5 * px + i] = cmid;
            }
            FDTD_HIP(hipMemcpy(xrow2_d, xr.data(), xr.size() * sizeof(T), hipMemcpyHostToDevice));
        }
        return FDTD_OK;
    }
    int set_array(int id, const double* host, size_t n) override {
        if (n != array_size(id) || (n && !host)) return fail(FDTD_ERR_ARG, "set_array(%d): expected %zu values, got %zu", id, array_size(id), n);
        if (id >= FDTD2D_TAB && id < FDTD2D_TAB + 24) {
            for (size_t q = 0; q < n; ++q) tab_h[(size_t)(id - FDTD2D_TAB) * kMaxLayers + q] = host[q];
            return upload_tables();
        }
        if (id == FDTD2D_JY) { for (size_t q = 0; q < n; ++q) Jy[q] = (T)host[q]; return FDTD_OK; }
        Map m;
        if (!make_map(id, m)) return fail(FDTD_ERR_ARG, "set_array: unknown array id %d", id);
        if (n == 0) return FDTD_OK;
        std::vector<T> st(m.dev_elems);
        FDTD_HIP(hipMemcpy(st.data(), m.dev, st.size() * sizeof(T), hipMemcpyDeviceToHost));
        size_t o = 0;
        for (int p0 = 0; p0 < m.n0; ++p0) for (int p1 = 0; p1 < m.n1; ++p1, ++o) {
            const long long off = dev_off(m, p0, p1);
            if (off >= 0) st[(size_t)off] = (T)host[o];
        }
        FDTD_HIP(hipMemcpy(m.dev, st.data(), st.size() * sizeof(T), hipMemcpyHostToDevice));
        if (m.dev2) FDTD_HIP(hipMemcpy(m.dev2, st.data(), st.size() * sizeof(T), hipMemcpyHostToDevice));
        pair_ready = false;               // the two-step kernel's spare buffers take the new values over on its next launch
        if (fused && id >= FDTD2D_C1 && id <= FDTD2D_C3) {
            const size_t tiles = (size_t)nz1 * ntx2;
            hipLaunchKernelGGL((k2_flags_fused<T, V>), dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, 0, C[0], C[1], C[2], disp2_d, px, nz1, ntx2, kLanes - 1);
            FDTD_HIP(hipGetLastError());
            const size_t tiles3 = (size_t)nz1 * ntx3;
            hipLaunchKernelGGL((k2_flags_fused<T, PV>), dim3((unsigned)((tiles3 + 3) / 4)), dim3(256), 0, 0, C[0], C[1], C[2], disp3_d, px, nz1, ntx3, kLanes - 2 * pair2d_halo_lanes<PV>());
            FDTD_HIP(hipGetLastError());
            std::vector<unsigned char> fl(tiles3);
            FDTD_HIP(hipMemcpy(fl.data(), disp3_d, tiles3, hipMemcpyDeviceToHost));
            if (int rc = pair_tile_lists(fl)) return rc;
        }
        if (id >= FDTD2D_C1 && id <= FDTD2D_C3) {
            const int ntx = px / (kLanes * V);
            const size_t tiles = (size_t)nz1 * ntx;
            hipLaunchKernelGGL((k2_flags<T, V>), dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, 0, C[0], C[1], C[2], disp_d, px, nz1, ntx);
            FDTD_HIP(hipGetLastError());
            FDTD_HIP(hipDeviceSynchronize());
        }
        return FDTD_OK;
    }
    int get_array(int id, double* host, size_t n) override {
        if (n != array_size(id) || (n && !host)) return fail(FDTD_ERR_ARG, "get_array(%d): expected %zu values, got %zu", id, array_size(id), n);
        if (id >= FDTD2D_TAB && id < FDTD2D_TAB + 24) { for (size_t q = 0; q < n; ++q) host[q] = tab_h[(size_t)(id - FDTD2D_TAB) * kMaxLayers + q]; return FDTD_OK; }
        if (id == FDTD2D_JY) { for (size_t q = 0; q < n; ++q) host[q] = (double)Jy[q]; return FDTD_OK; }
        Map m;
        if (!make_map(id, m)) return fail(FDTD_ERR_ARG, "get_array: unknown array id %d", id);
        if (n == 0) return FDTD_OK;
        const T* src = m.dev;
        if (fused && m.fam >= 0 && (m.slot == 1 || m.slot == 2)) { src = current_field(m.fam, 0); FDTD_HIP(hipGetLastError()); }
        FDTD_HIP(hipDeviceSynchronize());
        std::vector<T> st(m.dev_elems);
        FDTD_HIP(hipMemcpy(st.data(), src, st.size() * sizeof(T), hipMemcpyDeviceToHost));
        size_t o = 0;
        for (int p0 = 0; p0 < m.n0; ++p0) for (int p1 = 0; p1 < m.n1; ++p1, ++o) {
            const long long off = dev_off(m, p0, p1);
            host[o] = off >= 0 ? (double)st[(size_t)off] : 0.0;
        }
        return FDTD_OK;
    }
    int sumsq(int id, double* out, hipStream_t s) override {
        Map m;
        if (!make_map(id, m) || m.kind != 0) return fail(FDTD_ERR_ARG, "sumsq: array %d is not a field", id);
        const T* src = (fused && m.fam >= 0 && (m.slot == 1 || m.slot == 2)) ? current_field(m.fam, s) : m.dev;
        hipLaunchKernelGGL(k2_sumsq<T>, dim3(1024), dim3(256), 0, s, src, d.Nx, px, nz1, red_d);
        FDTD_HIP(hipGetLastError());
        std::vector<double> h(1024);
        FDTD_HIP(hipMemcpyAsync(h.data(), red_d, 1024 * sizeof(double), hipMemcpyDeviceToHost, s));
        FDTD_HIP(hipStreamSynchronize(s));
        double acc = 0;
        for (double v : h) acc += v;
        *out = acc;
        return FDTD_OK;
    }

    K2<T> args() {
        K2<T> a;
        a.Dy = Dy; a.Bx = Bx; a.Bz = Bz;
        for (int f = 0; f < 3; ++f) { a.F1[f] = Fa[f]; a.F2[f] = Fb[f]; a.S1[f] = Sa[f]; a.S2[f] = Sb[f]; }
        a.C1 = C[0]; a.C2 = C[1]; a.C3 = C[2];
        for (int s = 0; s < 2; ++s) { a.ph_ey_z[s] = ph_z[0][s]; a.ph_mx_z[s] = ph_z[1][s]; a.ph_ey_x[s] = ph_x[0][s]; a.ph_mz_x[s] = ph_x[1][s]; }
        a.tab = tab_d; a.xrow = xrow_d; a.disp = disp_d;
        a.Nx = d.Nx; a.Nz = d.Nz; a.px = px; a.nz1 = nz1;
        for (int ax = 0; ax < 2; ++ax) for (int s = 0; s < 2; ++s) a.np[ax][s] = np[ax][s];
        a.xwn = xwn; a.xwp = xwp; a.xp0 = xp0;
        a.iJ = d.i_Jy; a.kJ = d.k_Jy; a.src_on = 0; a.jy = (T)0;
        a.dt = (T)d.dt; a.dtd[0] = (T)(d.dt / d.dx); a.dtd[1] = (T)(d.dt / d.dz); a.mu0 = (T)d.mu0; a.eps0 = (T)d.eps0;
        a.chunk = chunk;
        a.remap = remap;
        a.ntx = px / (kLanes * V);
        a.nbx = (a.ntx + kWavesPerBlock - 1) / kWavesPerBlock;
        a.nbz = (nz1 + chunk - 1) / chunk;
        return a;
    }
    int step_fused(int n, hipStream_t s) {
        K2F<T> a;
        a.Dy = Dy; a.Bx = Bx; a.Bz = Bz; a.Dyn = Dy2; a.Bxn = Bx2; a.Bzn = Bz2;
        a.Ec = Fa[0]; a.Ep = Fb[0]; a.En = Ef; a.SEc = Sa[0]; a.SEp = Sb[0]; a.SEn = SEf;
        for (int f = 0; f < 2; ++f) { a.H1[f] = Fa[1 + f]; a.H2[f] = Fb[1 + f]; a.SH1[f] = Sa[1 + f]; a.SH2[f] = Sb[1 + f]; }
        a.C1 = C[0]; a.C2 = C[1]; a.C3 = C[2];
        for (int sd = 0; sd < 2; ++sd) {
            a.pe_z[sd] = ph_z[0][sd]; a.pe_zn[sd] = ph_z2[sd]; a.pe_x[sd] = ph_x[0][sd]; a.pe_xn[sd] = ph_x2[sd];
            a.pm_z[sd] = ph_z[1][sd]; a.pm_x[sd] = ph_x[1][sd];
        }
        a.tab = tab_d; a.xrow = xrow2_d; a.disp = disp2_d;
        a.Nx = d.Nx; a.Nz = d.Nz; a.px = px; a.nz1 = nz1;
        for (int ax = 0; ax < 2; ++ax) for (int sd = 0; sd < 2; ++sd) a.np[ax][sd] = np[ax][sd];
        a.xwn = xwn; a.xwp = xwp; a.xp0 = xp0;
        a.iJ = d.i_Jy; a.kJ = d.k_Jy;
        a.src_on = (n >= 0 && n < (int)Jy.size()) ? 1 : 0;
        a.dt = (T)d.dt; a.dtd[0] = (T)(d.dt / d.dx); a.dtd[1] = (T)(d.dt / d.dz); a.mu0 = (T)d.mu0; a.eps0 = (T)d.eps0;
        a.jy = a.src_on ? a.dt * Jy[n] : (T)0;
        a.chunk = chunk_f; a.remap = remap_set ? remap : 8; a.ntx = ntx2;     // runs of 8 tiles per XCD: 1.5 % over contiguous ranges at 8192^2
        a.nbx = ntx2;                                              // a workgroup = four consecutive row chunks of one x tile
        a.nbz = (nz1 + kStack2D * chunk_f - 1) / (kStack2D * chunk_f);
        timer.begin(2, s);
        hipLaunchKernelGGL(k2_fused<T>, dim3((unsigned)a.nbx * a.nbz), dim3(kLanes * kStack2D), 0, s, a);
        timer.end(s);
        FDTD_HIP(hipGetLastError());
        std::swap(Dy, Dy2); std::swap(Bx, Bx2); std::swap(Bz, Bz2);
        for (int sd = 0; sd < 2; ++sd) { std::swap(ph_z[0][sd], ph_z2[sd]); std::swap(ph_x[0][sd], ph_x2[sd]); }
        { T* t = Fb[0]; Fb[0] = Fa[0]; Fa[0] = Ef; Ef = t; }            // new -> n-1 -> n-2 -> free
        { T* t = Sb[0]; Sb[0] = Sa[0]; Sa[0] = SEf; SEf = t; }
        for (int f = 1; f < 3; ++f) { std::swap(Fa[f], Fb[f]); std::swap(Sa[f], Sb[f]); }
        return probe_record(s);
    }
    // ---- two steps per launch
    // x tiles with and without Lorentz cells (any row): each kind gets a launch of its own kernel variant
    int* tiles3_d = nullptr;
    int n_plain3 = 0, n_lz3 = 0;
    ConstDiv<T> cd_mu{}, cd_eps{};
    hipStream_t lz_stream = nullptr;
    hipEvent_t lz_fork = nullptr, lz_join = nullptr;
    int pair_tile_lists(const std::vector<unsigned char>& flags) {
        std::vector<int> plain, lz;
        for (int t = 0; t < ntx3; ++t) {
            bool any = false;
            for (int k = 0; k < nz1 && !any; ++k) any = !flags.empty() && flags[(size_t)k * ntx3 + t] != 0;
            (any ? lz : plain).push_back(t);
        }
        n_plain3 = (int)plain.size(); n_lz3 = (int)lz.size();
        any_disp = n_lz3 > 0;
        plain.insert(plain.end(), lz.begin(), lz.end());
        if (!tiles3_d) if (int rc = dalloc((void**)&tiles3_d, (size_t)ntx3 * sizeof(int) + 64)) return rc;
        FDTD_HIP(hipMemcpy(tiles3_d, plain.data(), plain.size() * sizeof(int), hipMemcpyHostToDevice));
        return FDTD_OK;
    }
    // Cells the kernels never write (rows and columns outside the sweeps, samples and sums of cells without Lorentz
    // coefficients) must read the same from whichever buffer is current, so the spare buffers start as copies.
    int pair_prepare() {
        int rc;
        if (!pair_allocated) {
            for (int sd = 0; sd < 2; ++sd) {
                if (np[1][sd] && (rc = dalloc((void**)&pm_z2[sd], ((size_t)np[1][sd] * px + px) * sizeof(T)))) return rc;
                if (np[0][sd] && (rc = dalloc((void**)&pm_x2[sd], ((size_t)nz1 * (sd ? xwp : xwn) + px) * sizeof(T)))) return rc;
            }
            pair_allocated = true;
        }
        if (any_disp && !Fx[0]) {
            const size_t fb = (fsize + px) * sizeof(T);
            for (int f = 0; f < 3; ++f)
                if ((rc = dalloc((void**)&Fx[f], fb)) || (rc = dalloc((void**)&Fy[f], fb)) || (rc = dalloc((void**)&Sx[f], fb)) || (rc = dalloc((void**)&Sy[f], fb))) return rc;
        }
        for (int sd = 0; sd < 2; ++sd) {
            if (pm_z2[sd]) FDTD_HIP(hipMemcpy(pm_z2[sd], ph_z[1][sd], (size_t)np[1][sd] * px * sizeof(T), hipMemcpyDeviceToDevice));
            if (pm_x2[sd]) FDTD_HIP(hipMemcpy(pm_x2[sd], ph_x[1][sd], (size_t)nz1 * (sd ? xwp : xwn) * sizeof(T), hipMemcpyDeviceToDevice));
        }
        if (Fx[0])
            for (int f = 0; f < 3; ++f) {
                FDTD_HIP(hipMemcpy(Fy[f], Fa[f], fsize * sizeof(T), hipMemcpyDeviceToDevice));
                FDTD_HIP(hipMemcpy(Fx[f], Fb[f], fsize * sizeof(T), hipMemcpyDeviceToDevice));
                FDTD_HIP(hipMemcpy(Sy[f], Sa[f], fsize * sizeof(T), hipMemcpyDeviceToDevice));
                FDTD_HIP(hipMemcpy(Sx[f], Sb[f], fsize * sizeof(T), hipMemcpyDeviceToDevice));
            }
        pair_ready = true;
        return FDTD_OK;
    }
    // whether step() takes its steps two per launch: never with probes (a probe is sampled after every step); by default
    // on grids of at least 4 Mi cells (smaller ones do not fill the chip with the two-step kernel's longer waves);
    // FDTD2D_PAIR=0/1 decides otherwise
    bool pair_wanted() const {
        if (!fused || !probes.specs.empty() || pair_env == 0) return false;
        return pair_env == 1 || fsize >= ((size_t)4 << 20);
    }
    int step_pair(int n, hipStream_t s) {
        if (!pair_ready) { FDTD_HIP(hipStreamSynchronize(s)); if (int rc = pair_prepare()) return rc; }
        K2P<T> a;
        a.Dy = Dy; a.Bx = Bx; a.Bz = Bz; a.Dyn = Dy2; a.Bxn = Bx2; a.Bzn = Bz2;
        for (int f = 0; f < 3; ++f) {
            a.F1[f] = Fa[f]; a.F2[f] = Fb[f]; a.FX[f] = Fx[f]; a.FY[f] = Fy[f];
            a.S1[f] = Sa[f]; a.S2[f] = Sb[f]; a.SX[f] = Sx[f]; a.SY[f] = Sy[f];
        }
        a.C1 = C[0]; a.C2 = C[1]; a.C3 = C[2];
        for (int sd = 0; sd < 2; ++sd) {
            a.pe_z[sd] = ph_z[0][sd]; a.pe_zn[sd] = ph_z2[sd]; a.pe_x[sd] = ph_x[0][sd]; a.pe_xn[sd] = ph_x2[sd];
            a.pm_z[sd] = ph_z[1][sd]; a.pm_zn[sd] = pm_z2[sd]; a.pm_x[sd] = ph_x[1][sd]; a.pm_xn[sd] = pm_x2[sd];
        }
        a.tab = tab_d; a.xrow = xrow2_d; a.disp = disp3_d;
        a.Nx = d.Nx; a.Nz = d.Nz; a.px = px; a.nz1 = nz1;
        for (int ax = 0; ax < 2; ++ax) for (int sd = 0; sd < 2; ++sd) a.np[ax][sd] = np[ax][sd];
        a.xwn = xwn; a.xwp = xwp; a.xp0 = xp0;
        a.iJ = d.i_Jy; a.kJ = d.k_Jy;
        const bool s1 = n >= 0 && n < (int)Jy.size(), s2 = n + 1 >= 0 && n + 1 < (int)Jy.size();
        a.src_on = (s1 ? 1 : 0) | (s2 ? 2 : 0);
        a.dt = (T)d.dt; a.dtd[0] = (T)(d.dt / d.dx); a.dtd[1] = (T)(d.dt / d.dz); a.mu0 = (T)d.mu0; a.eps0 = (T)d.eps0;
        a.jy = s1 ? a.dt * Jy[n] : (T)0;
        a.jy2 = s2 ? a.dt * Jy[n + 1] : (T)0;
        a.dm = cd_mu; a.de = cd_eps;
        const bool acc = pair_acc_flags >= 0;
        a.acc = acc ? obs_acc : nullptr; a.acc_snap = acc ? obs_snap : nullptr; a.acc_w = pair_acc_w; a.acc_flags = acc ? pair_acc_flags : 0;
        a.chunk = chunk_p; a.ntx = ntx3;
        a.nbz = (nz1 + chunk_p - 1) / chunk_p;
        a.remap = remap_set ? remap : 0;
        if (!tiles3_d) if (int rc = pair_tile_lists({})) return rc;       // (no coefficient was ever uploaded: no Lorentz cell)
        // the tiles with Lorentz cells (usually few) run beside the others, on a stream of their own
        const bool beside = n_plain3 > 0 && n_lz3 > 0;
        if (beside && !lz_stream) {
            FDTD_HIP(hipStreamCreateWithFlags(&lz_stream, hipStreamNonBlocking));
            FDTD_HIP(hipEventCreateWithFlags(&lz_fork, hipEventDisableTiming));
            FDTD_HIP(hipEventCreateWithFlags(&lz_join, hipEventDisableTiming));
        }
        timer.begin(3, s);
        if (n_lz3) {
            hipStream_t ls = beside ? lz_stream : s;
            if (beside) { FDTD_HIP(hipEventRecord(lz_fork, s)); FDTD_HIP(hipStreamWaitEvent(ls, lz_fork, 0)); }
            K2P<T> b = a;
            b.tiles = tiles3_d + n_plain3; b.n_tiles = n_lz3; b.nbx = (n_lz3 + kPairTiles2D - 1) / kPairTiles2D;
            if (acc) hipLaunchKernelGGL((k2_pair<T, PV, true, true>), dim3((unsigned)b.nbx * b.nbz), dim3(kLanes * kPairTiles2D), 0, ls, b);
            else hipLaunchKernelGGL((k2_pair<T, PV, true>), dim3((unsigned)b.nbx * b.nbz), dim3(kLanes * kPairTiles2D), 0, ls, b);
            if (beside) FDTD_HIP(hipEventRecord(lz_join, ls));
        }
        if (n_plain3) {
            a.tiles = tiles3_d; a.n_tiles = n_plain3; a.nbx = (n_plain3 + kPairTiles2D - 1) / kPairTiles2D;
            if (acc) hipLaunchKernelGGL((k2_pair<T, PV, false, true>), dim3((unsigned)a.nbx * a.nbz), dim3(kLanes * kPairTiles2D), 0, s, a);
            else hipLaunchKernelGGL((k2_pair<T, PV, false>), dim3((unsigned)a.nbx * a.nbz), dim3(kLanes * kPairTiles2D), 0, s, a);
        }
        if (beside) FDTD_HIP(hipStreamWaitEvent(s, lz_join, 0));
        timer.end(s);
        FDTD_HIP(hipGetLastError());
        std::swap(Dy, Dy2); std::swap(Bx, Bx2); std::swap(Bz, Bz2);
        for (int sd = 0; sd < 2; ++sd) {
            std::swap(ph_z[0][sd], ph_z2[sd]); std::swap(ph_x[0][sd], ph_x2[sd]);
            std::swap(ph_z[1][sd], pm_z2[sd]); std::swap(ph_x[1][sd], pm_x2[sd]);
        }
        if (Fx[0])
            for (int f = 0; f < 3; ++f) {       // after the second step -> n-1, after the first -> n-2; the old pair is the next launch's spare
                std::swap(Fa[f], Fy[f]); std::swap(Fb[f], Fx[f]);
                std::swap(Sa[f], Sy[f]); std::swap(Sb[f], Sx[f]);
            }
        return FDTD_OK;
    }
    int step(int n0, int count, hipStream_t s) override {
        if (fused) {
            int n = n0;
            if (pair_wanted())
                for (; n + 1 < n0 + count; n += 2) if (int rc = step_pair(n, s)) return rc;
            for (; n < n0 + count; ++n) if (int rc = step_fused(n, s)) return rc;
            return FDTD_OK;
        }
        for (int n = n0; n < n0 + count; ++n) {
            K2<T> a = args();
            a.src_on = (n >= 0 && n < (int)Jy.size()) ? 1 : 0;
            a.jy = a.src_on ? a.dt * Jy[n] : (T)0;
            const unsigned nblk = (unsigned)a.nbx * a.nbz;
            timer.begin(0, s);
            hipLaunchKernelGGL(k2_part_e<T>, dim3(nblk), dim3(256), 0, s, a);
            timer.end(s);
            FDTD_HIP(hipGetLastError());
            std::swap(Fa[0], Fb[0]); std::swap(Sa[0], Sb[0]);       // new Ey/SEy now sit in the "current" slot
            a = args();
            timer.begin(1, s);
            hipLaunchKernelGGL(k2_part_h<T>, dim3(nblk), dim3(256), 0, s, a);
            timer.end(s);
            FDTD_HIP(hipGetLastError());
            for (int f = 1; f < 3; ++f) { std::swap(Fa[f], Fb[f]); std::swap(Sa[f], Sb[f]); }
            if (int rc = probe_record(s)) return rc;
        }
        return FDTD_OK;
    }
};

}  // namespace

extern "C" {

int fdtd2d_create(const fdtd2d_desc* d, fdtd2d_handle* out) {
    if (!out) return fail(FDTD_ERR_ARG, "null output handle");
    *out = nullptr;
    if (!d) return fail(FDTD_ERR_ARG, "null descriptor");
    if (d->struct_size != (int32_t)sizeof(fdtd2d_desc)) return fail(FDTD_ERR_ARG, "descriptor size %d, library expects %zu", d->struct_size, sizeof(fdtd2d_desc));
    if (d->precision != 32 && d->precision != 64) return fail(FDTD_ERR_ARG, "precision must be 32 or 64");
    if (d->mode != FDTD2D_MODE_FULL && d->mode != FDTD2D_MODE_FUSED) return fail(FDTD_ERR_ARG, "unknown 2-D mode %d", d->mode);
    if (d->Nx < 2 || d->Nz < 2) return fail(FDTD_ERR_ARG, "grid too small");
    for (int q = 0; q < 4; ++q) if (d->n_pml[q] < 0 || d->n_pml[q] > kMaxLayers) return fail(FDTD_ERR_ARG, "CPML layer counts must be in [0,%d]", kMaxLayers);
    if (d->n_pml[0] + d->n_pml[1] >= d->Nx || d->n_pml[2] + d->n_pml[3] >= d->Nz) return fail(FDTD_ERR_ARG, "CPML layers do not fit the grid");
    // the reference silently skips a source outside [0,Nx]x[0,Nz] (FDTD2D.cpp:360); column Nx is the
    // wall column this engine does not store, so a source there is refused instead of dropped
    if (d->i_Jy == d->Nx) return fail(FDTD_ERR_ARG, "source column i_Jy = Nx lies on the wall");
    if (!(d->dx > 0 && d->dz > 0 && d->dt > 0 && d->mu0 > 0 && d->eps0 > 0)) return fail(FDTD_ERR_ARG, "dx, dz, dt, mu0, eps0 must be positive");
    int ndev = 0;
    FDTD_HIP(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(FDTD_ERR_HIP, "no HIP device visible");
    if (d->device >= ndev) return fail(FDTD_ERR_ARG, "device %d requested, %d visible", d->device, ndev);
    try {
        DeviceScope scope(d->device);
        if (scope.status != hipSuccess) return fail(FDTD_ERR_HIP, "cannot select device %d: %s", d->device, hipGetErrorString(scope.status));
        int dev = 0, rc;
        FDTD_HIP(hipGetDevice(&dev));
        if (d->precision == 32) { auto e = std::make_unique<Engine2<float>>(); e->device = dev; if ((rc = e->init(*d))) return rc; *out = e.release(); }
        else { auto e = std::make_unique<Engine2<double>>(); e->device = dev; if ((rc = e->init(*d))) return rc; *out = e.release(); }
    } catch (const std::bad_alloc&) {
        return fail(FDTD_ERR_HOST, "out of host memory");
    }
    return FDTD_OK;
}
int fdtd2d_destroy(fdtd2d_handle h) {
    if (!h) return FDTD_OK;
    DeviceScope scope(h->device);
    delete h;
    return FDTD_OK;
}
size_t fdtd2d_array_size(fdtd2d_handle h, int id) { return h ? h->array_size(id) : 0; }
int fdtd2d_set_array(fdtd2d_handle h, int id, const double* host, size_t n) { return entry(h, [&] { return h->set_array(id, host, n); }); }
int fdtd2d_get_array(fdtd2d_handle h, int id, double* host, size_t n) { return entry(h, [&] { return h->get_array(id, host, n); }); }
int fdtd2d_step(fdtd2d_handle h, int n0, int count, void* stream) { return entry(h, [&] { return h->step(n0, count, (hipStream_t)stream); }); }
int fdtd2d_sumsq(fdtd2d_handle h, int id, double* out, void* stream) {
    if (!out) return fail(FDTD_ERR_ARG, "null argument");
    return entry(h, [&] { return h->sumsq(id, out, (hipStream_t)stream); });
}
size_t fdtd2d_device_bytes(fdtd2d_handle h) { return h ? h->device_bytes() : 0; }
int fdtd2d_observe_accumulate(fdtd2d_handle h, double weight, int reset, void* stream) { return entry(h, [&] { return h->observe_accumulate(weight, reset, (hipStream_t)stream); }); }
int fdtd2d_observe_rms(fdtd2d_handle h, double* host, size_t n) { return entry(h, [&] { return h->observe_rms(host, n); }); }
int fdtd2d_step_observed(fdtd2d_handle h, int n0, int count, double weight, int window, void* stream) {
    return entry(h, [&] { return h->step_observed(n0, count, weight, window, (hipStream_t)stream); });
}
int fdtd2d_observe_rms_window(fdtd2d_handle h, double* host, size_t n) { return entry(h, [&] { return h->observe_rms_window(host, n); }); }
int fdtd_const_division_proven(double c) {
    fdtd::ConstDiv<float> d{};
    if (make_const_div<float>(c, d) != FDTD_OK) return -1;
    return d.fast;
}
// (test hook, not in the header: disagreements between the short and the IEEE division over every fp32 numerator with
// |x| in [lo, hi] or x = +0 -- the proof above with bounds of the caller's choice; -1 on error)
long long fdtd_const_division_mismatches(double c, double lo, double hi) {
    fdtd::ConstDiv<float> d{(float)c, 1.0f / (float)c, (float)lo, (float)hi, 1};
    unsigned long long* bad_d = nullptr;
    unsigned long long bad = 0;
    if (hipMalloc((void**)&bad_d, sizeof bad) != hipSuccess) return -1;
    (void)hipMemset(bad_d, 0, sizeof bad);
    hipLaunchKernelGGL(fdtd::k_const_div_verify, dim3(256 * 64), dim3(256), 0, 0, d, bad_d);
    const bool ok = hipMemcpy(&bad, bad_d, sizeof bad, hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(bad_d);
    return ok ? (long long)bad : -1;
}
int fdtd2d_sync(fdtd2d_handle h, void* stream) {
    return entry(h, [&] { FDTD_HIP(hipStreamSynchronize((hipStream_t)stream)); return (int)FDTD_OK; });
}
int fdtd2d_probe_add(fdtd2d_handle h, int id, int i, int k) { return entry(h, [&] { return h->probe_add(id, i, k); }); }
int fdtd2d_probe_record(fdtd2d_handle h, void* stream) { return entry(h, [&] { return h->probe_record((hipStream_t)stream); }); }
int fdtd2d_probe_read(fdtd2d_handle h, double* out, size_t cap, int* n_samples, int* n_probes) { return entry(h, [&] { return h->probe_read(out, cap, n_samples, n_probes); }); }
int fdtd2d_probe_reset(fdtd2d_handle h, int drop_probes) { return entry(h, [&] { return h->probe_reset(drop_probes); }); }
int fdtd2d_kernel_timing(fdtd2d_handle h, int enable) { if (!h) return fail(FDTD_ERR_ARG, "null handle"); h->timer.enabled = enable != 0; return FDTD_OK; }
int fdtd2d_kernel_time(fdtd2d_handle h, int kind, int* launches, double* total_ms) { return entry(h, [&] { return h->timer.collect(kind, launches, total_ms); }); }

}  // extern "C"
