// kernels3d.hpp -- gfx950 kernels of the 3-D Yee update (E-half and H-half).
//
// Replaces FDTD3D::updateE / applySource / updateElectricMaterialRelations and
// FDTD3D::updateH / updateMagneticMaterialRelations (fdtd3d/src/FDTD3D.cpp:480-1021).
//
// Device layout of every field component: [plane kl][row j][i], i unit-stride, rows padded to a
// multiple of 256 elements, Ny+1 rows, (owned layers + 2) planes -- plane 0 is the ghost layer
// below the slab, the last plane the ghost layer above it (or the closing wall layer k = Nz).
// Components whose x extent is Nx+1 do not store i = Nx: neither half step ever writes that
// wall column (all sweeps over those components stop at Nx-1), so it is identically zero.
//
// Work decomposition: a 256-thread workgroup owns an x tile of 64 lanes x 16 bytes and four
// consecutive rows (one wave per row) and marches through a chunk of planes.  The z-neighbour
// needed by the z derivative travels in registers from one plane iteration to the next; the
// y-neighbour row is the row a sibling wave is loading at the same moment (L1/L2 hit); the
// x-neighbour is one cross-lane shift.  Every field word is therefore fetched from HBM once
// per half step.  All three axis terms, the source and the constitutive division are applied
// in one pass, in the reference's per-cell order of floating-point operations.
#pragma once
#include "common.hpp"

namespace fdtd {

template <typename T>
struct K3 {
    T* D[3];
    T* E[3];
    T* B[3];
    T* H[3];
    const T* eps[3];
    const T* mu[3];
    T* phE[3][3][2];   // [component][axis][side]
    T* phM[3][3][2];
    const T* tab;      // [axis][side][kind 6][kMaxLayers]
    const T* xrow;     // [kind 6][px]: x-face coefficients expanded per i
    const T* mag;      // [ny1][px]
    int Nx, Ny, Nz;    // global cell counts
    int px, ny1, nzp;  // row pitch, rows per plane, planes stored
    int zoff;          // global k = kl + zoff
    int np[3][2];
    int xwn, xwp, xp0; // x-face auxiliary strips: widths and first column of the high strip
    int kJ;            // global source plane
    int src_on;
    T jx;              // dt * wf[n]
    T dt, dtd[3], mu0;
    int kl0, kl1;      // planes to process [kl0, kl1)
    int chunk;         // planes per workgroup
    int nbx, nby, nbz;
    int remap;         // 1: hand each XCD a contiguous range of tiles
    int wpb;           // waves (= rows) per workgroup
    // fused whole-step kernel only: the buffers this step writes (D, B and the D-side CPML state ping-pong)
    T* Dn[3];
    T* Bn[3];
    T* phEn[3][3][2];
    int kl_stop;       // planes below this index may be visited for the closing D iteration of a chunk
    int copy_top;      // plane kl_stop-1 is a top layer whose B is finished later: carry it over unchanged
    // two-step kernel only: dt * wf[n+1]; src_on bit 0 = the first step has a source term, bit 1 = the second
    T jx2;
    T* phMn[3][3][2];  // the B-side CPML state it writes (ping-pong, like phEn)
    int zcut[2], zchunk[3], znb[2];   // the two-step kernel's three plane segments (kernels3d_pair.hpp)
    T* pair_scr;       // CPML state between the two steps of a launch: [workgroup][wave][2 planes][12 terms][columns of a wave]
    // two-step kernel: x tile 0 may hold BOTH ends of a row (lanes [0, wrap_l) the first columns, the other lanes the
    // columns from wrap_r0 on), so that one x tile instead of two meets the x-face CPML strips; 0 = tiles in a row
    int wrap_l, wrap_r0;
    int peek;          // one-step fused kernel: leave the B-side CPML state (updated in place) as it is -- a launch whose results are only sampled
    // two-step kernel with the RMS window folded in: the window [plane][row][column] like a field array, the array that
    // keeps a window as the launch's FIRST step leaves it, the weight, and flags (1: the first step opens a new window,
    // 2: the second step does, 4: keep the window after the first step)
    T* acc;
    T* acc_snap;
    T acc_w;
    int acc_flags;
    int acc_defer_kl;  // local plane of a cell layer whose SECOND-step contribution the launch leaves out (-1: none): the top layer of a
                       // slab below another one, whose second-step E above it is only exact after the planes beyond the cut have arrived
};

template <typename T, int V>
__device__ __forceinline__ T left_of(const Pack<T, V>& x, const T* p, int i0, int lane) {
    T l = lane_below(x.v[V - 1]);
    if (lane == 0) l = (i0 > 0) ? p[-1] : (T)0;
    return l;
}
template <typename T, int V>
__device__ __forceinline__ T right_of(const Pack<T, V>& x, const T* p, int i0, int lane, int px) {
    T r = lane_above(x.v[0]);
    if (lane == kLanes - 1) r = (i0 + V < px) ? p[V] : (T)0;
    return r;
}
template <typename T, int V>
__device__ __forceinline__ Pack<T, V> diff(const Pack<T, V>& a, const Pack<T, V>& b) {
    Pack<T, V> r;
#pragma unroll
    for (int e = 0; e < V; ++e) r.v[e] = a.v[e] - b.v[e];
    return r;
}
// a[e] - a[e-1]
template <typename T, int V>
__device__ __forceinline__ Pack<T, V> diff_back_x(const Pack<T, V>& a, T left) {
    Pack<T, V> r;
    r.v[0] = a.v[0] - left;
#pragma unroll
    for (int e = 1; e < V; ++e) r.v[e] = a.v[e] - a.v[e - 1];
    return r;
}
// a[e+1] - a[e]
template <typename T, int V>
__device__ __forceinline__ Pack<T, V> diff_fwd_x(const Pack<T, V>& a, T right) {
    Pack<T, V> r;
#pragma unroll
    for (int e = 0; e < V - 1; ++e) r.v[e] = a.v[e + 1] - a.v[e];
    r.v[V - 1] = right - a.v[V - 1];
    return r;
}

// Plain update: d +- (dt/d) * curl   (FDTD3D.cpp:572-573 pattern)
template <int SIGN, typename T, int V>
__device__ __forceinline__ void apply_mid(Pack<T, V>& d, const Pack<T, V>& curl, unsigned mask, T c) {
#pragma unroll
    for (int e = 0; e < V; ++e)
        if (mask & (1u << e)) {
            const T v = c * curl.v[e];
            d.v[e] = SIGN > 0 ? d.v[e] + v : d.v[e] - v;
        }
}
// CPML layer: phi = b*phi + a*curl;  d +- (C*curl + dt*phi)   (FDTD3D.cpp:545-548 pattern)
template <int SIGN, typename T, int V>
__device__ __forceinline__ void apply_pml(Pack<T, V>& d, const Pack<T, V>& curl, unsigned mask, T b, T a, T C, T dt, T* phi) {
    Pack<T, V> ph = ldv<T, V>(phi);
#pragma unroll
    for (int e = 0; e < V; ++e)
        if (mask & (1u << e)) {
            const T p = b * ph.v[e] + a * curl.v[e];
            ph.v[e] = p;
            const T v = C * curl.v[e] + dt * p;
            d.v[e] = SIGN > 0 ? d.v[e] + v : d.v[e] - v;
        }
    stv<T, V>(phi, ph);
}
// Term along an axis whose zone is the same for the whole wave (y: one row per wave, z: one
// plane per iteration).  `face` points at the six tables of side 0 of that axis.
template <int SIGN, typename T, int V>
__device__ __forceinline__ void apply_uniform(Pack<T, V>& d, const Pack<T, V>& curl, unsigned mask, int zone, int l,
                                              const T* __restrict__ face, int kb, T cmid, T dt, T* phi_n, T* phi_p) {
    if (mask == 0u || zone == Z_NONE) return;
    if (zone == Z_MID) { apply_mid<SIGN, T, V>(d, curl, mask, cmid); return; }
    // (tables through the scalar cache: the layer is the same for the whole wave)
    const FDTD_AS(4) T* t = (const FDTD_AS(4) T*)(face + (zone == Z_POS ? 6 * kMaxLayers : 0) + kb * kMaxLayers + l);
    apply_pml<SIGN, T, V>(d, curl, mask, t[0], t[kMaxLayers], t[2 * kMaxLayers], dt, zone == Z_POS ? phi_p : phi_n);
}
// Term along x: the zone changes from element to element.  Coefficients come from rows expanded
// per column; the auxiliary strips are only touched by lanes that overlap them.
// OPEN_SKIP selects the 3-D forward-difference rule that leaves the last cell alone when the high
// face has no CPML (zone_h); the 2-D solver does not have it.
template <int SIGN, bool HTYPE, bool OPEN_SKIP, typename T, int V>
__device__ __forceinline__ void apply_x(Pack<T, V>& d, const Pack<T, V>& curl, unsigned mask, int i0, int Nx, int nn, int np,
                                        const T* __restrict__ xrow, int px, T cmid, T dt, T* phi_n, T* phi_p) {
    if (mask == 0u) return;
    const bool on_n = phi_n != nullptr, on_p = phi_p != nullptr;
    Pack<T, V> pn = zero_pack<T, V>(), pp = zero_pack<T, V>(), b = zero_pack<T, V>(), a = zero_pack<T, V>(), C = zero_pack<T, V>();
    if (on_n || on_p) {
        const T* r = xrow + (HTYPE ? 3 : 0) * (size_t)px + i0;
        b = ldv<T, V>(r); a = ldv<T, V>(r + px); C = ldv<T, V>(r + 2 * (size_t)px);
        if (on_n) pn = ldv<T, V>(phi_n);
        if (on_p) pp = ldv<T, V>(phi_p);
    }
#pragma unroll
    for (int e = 0; e < V; ++e)
        if (mask & (1u << e)) {
            int l;
            const int z = HTYPE ? zone_h(i0 + e, Nx, nn, np, OPEN_SKIP, l) : zone_e(i0 + e, Nx, nn, np, l);
            if (z == Z_MID) {
                const T v = cmid * curl.v[e];
                d.v[e] = SIGN > 0 ? d.v[e] + v : d.v[e] - v;
            } else if (z == Z_NEG || z == Z_POS) {
                const T old = (z == Z_NEG) ? pn.v[e] : pp.v[e];
                const T p = b.v[e] * old + a.v[e] * curl.v[e];
                if (z == Z_NEG) pn.v[e] = p; else pp.v[e] = p;
                const T v = C.v[e] * curl.v[e] + dt * p;
                d.v[e] = SIGN > 0 ? d.v[e] + v : d.v[e] - v;
            }
        }
    if (on_n) stv<T, V>(phi_n, pn);
    if (on_p) stv<T, V>(phi_p, pp);
}

// Field outputs of a half step are not read again before the next kernel: stream them past the caches
// (about 1.5 % with all of D, E, B, H resident).
template <typename T, int V>
__device__ __forceinline__ void stv_nt(T* p, const Pack<T, V>& x) {
    typedef T vec __attribute__((ext_vector_type(V)));
    vec v;
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = x.v[e];
    __builtin_nontemporal_store(v, reinterpret_cast<vec*>(p));
}
template <typename T, int V>
__device__ __forceinline__ Pack<T, V> divide(const Pack<T, V>& n, const Pack<T, V>& d) {
    Pack<T, V> r;
#pragma unroll
    for (int e = 0; e < V; ++e) r.v[e] = n.v[e] / d.v[e];
    return r;
}
template <typename T, int V>
__device__ __forceinline__ Pack<T, V> divide(const Pack<T, V>& n, T d) {
    Pack<T, V> r;
#pragma unroll
    for (int e = 0; e < V; ++e) r.v[e] = n.v[e] / d;
    return r;
}

// ---- division by a constant ------------------------------------------------------------------
// x / c for a wave-uniform c is, for most x, q0 = x*r, q = fma(fma(-q0, c, x), r, q0) with r = 1/c rounded: three
// instructions instead of the eleven of the hardware's IEEE sequence.  "Most" is made exact by the host: when an engine
// is created it runs EVERY fp32 bit pattern x of the class below through both and only sets `fast` if all quotients
// agree bit for bit (fdtd_const_division_proven).  The class: |x| in [lo, hi], or x = +0.  A wave takes the short
// sequence when all its numerators are in the class, the IEEE division otherwise (NaN, infinities, denormals, -0,
// huge and tiny values), so the result is the IEEE quotient always.  fp64 always divides.
template <typename T>
struct ConstDiv { T c, r, lo, hi; int fast; };

__device__ __forceinline__ bool const_div_in_class(float x, const ConstDiv<float>& d) {
    const float m = __builtin_fabsf(x);
    return m <= d.hi && (m >= d.lo || __builtin_amdgcn_classf(x, 0x040));       // 0x040: +0
}
__device__ __forceinline__ float const_div_short(float x, const ConstDiv<float>& d) {
    const float q0 = x * d.r;
    return __builtin_fmaf(__builtin_fmaf(-q0, d.c, x), d.r, q0);
}
template <typename T, int V>
__device__ __forceinline__ Pack<T, V> divide(const Pack<T, V>& n, const ConstDiv<T>& d) {
    if constexpr (sizeof(T) == 4) {
        if (d.fast) {
            bool ok = true;
#pragma unroll
            for (int e = 0; e < V; ++e) ok = ok && const_div_in_class(n.v[e], d);
            if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) {
                Pack<T, V> r;
#pragma unroll
                for (int e = 0; e < V; ++e) r.v[e] = const_div_short(n.v[e], d);
                return r;
            }
        }
    }
    return divide<T, V>(n, d.c);
}
// two packs with one decision
template <typename T, int V>
__device__ __forceinline__ void divide2(const Pack<T, V>& n0, const Pack<T, V>& n1, const ConstDiv<T>& d, Pack<T, V>& q0, Pack<T, V>& q1) {
    if constexpr (sizeof(T) == 4) {
        if (d.fast) {
            bool ok = true;
#pragma unroll
            for (int e = 0; e < V; ++e) ok = ok && const_div_in_class(n0.v[e], d) && const_div_in_class(n1.v[e], d);
            if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) {
#pragma unroll
                for (int e = 0; e < V; ++e) { q0.v[e] = const_div_short(n0.v[e], d); q1.v[e] = const_div_short(n1.v[e], d); }
                return;
            }
        }
    }
    q0 = divide<T, V>(n0, d.c); q1 = divide<T, V>(n1, d.c);
}

// ---- loop-invariant per-lane state of the 3-D kernels ---------------------------------------
template <typename T, int V>
struct Lane3 {
    int lane, i0, j, klb, kle;
    bool ok;
    bool first;                 // this lane's pack starts at the wall column i = 0
    bool in_n, in_p;            // pack lies in the low / high x auxiliary strip
    Pack<T, V> xb, xa, xC;      // x-axis coefficients of this lane's columns (see upload_tables)
};
template <typename T, int V, bool HTYPE>
__device__ __forceinline__ Lane3<T, V> decode_lane(const K3<T>& a) {
    Lane3<T, V> t;
    t.lane = threadIdx.x & (kLanes - 1);
    const int w = threadIdx.x >> 6;
    unsigned bid = a.remap == 1 ? xcd_contiguous(blockIdx.x, gridDim.x) : (a.remap > 1 ? xcd_grouped(blockIdx.x, gridDim.x, (unsigned)a.remap) : blockIdx.x);
    const int bx = bid % a.nbx; bid /= a.nbx;
    const int by = bid % a.nby;
    const int bz = bid / a.nby;
    t.i0 = (bx * kLanes + t.lane) * V;
    t.j = by * a.wpb + w;
    t.klb = a.kl0 + bz * a.chunk;
    t.kle = min(t.klb + a.chunk, a.kl1);
    t.ok = t.j < a.ny1;
    t.first = t.i0 == 0;
    t.in_n = a.np[0][0] > 0 && t.i0 < a.xwn;
    t.in_p = a.np[0][1] > 0 && t.i0 >= a.xp0 && t.i0 < a.xp0 + a.xwp;
    const T* r = a.xrow + (HTYPE ? 3 : 0) * (size_t)a.px + t.i0;
    t.xb = ldv<T, V>(r); t.xa = ldv<T, V>(r + a.px); t.xC = ldv<T, V>(r + 2 * (size_t)a.px);
    return t;
}

// Term along y or z: one zone for the whole wave.  Padding columns (i >= Nx) hold zeros in every
// input and so stay zero; `skip0` leaves the wall column i = 0 untouched for the terms whose sweeps
// start at i = 1.  Same operation order per cell as apply_mid / apply_pml above.
template <int SIGN, typename T, int V>
__device__ __forceinline__ void term_u(Pack<T, V>& d, const Pack<T, V>& curl, bool skip0, int zone, int l,
                                       const T* __restrict__ face, int kb, T cmid, T dt, T* phi_n, T* phi_p) {
    if (zone == Z_NONE) return;
    const T d0 = d.v[0];
    if (zone == Z_MID) {
#pragma unroll
        for (int e = 0; e < V; ++e) { const T v = cmid * curl.v[e]; d.v[e] = SIGN > 0 ? d.v[e] + v : d.v[e] - v; }
    } else {
        const FDTD_AS(4) T* t = (const FDTD_AS(4) T*)(face + (zone == Z_POS ? 6 * kMaxLayers : 0) + kb * kMaxLayers + l);
        const T b = t[0], a = t[kMaxLayers], C = t[2 * kMaxLayers];
        T* phi = zone == Z_POS ? phi_p : phi_n;
        Pack<T, V> ph = ldv<T, V>(phi);
        const T p0 = ph.v[0];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const T p = b * ph.v[e] + a * curl.v[e];
            ph.v[e] = p;
            const T v = C * curl.v[e] + dt * p;
            d.v[e] = SIGN > 0 ? d.v[e] + v : d.v[e] - v;
        }
        if (skip0) ph.v[0] = p0;
        stv<T, V>(phi, ph);
    }
    if (skip0) d.v[0] = d0;
}
// Term along x.  Every column carries its own coefficients (b, a, C): CPML columns the tabulated
// ones, plain columns (0, 0, dt/dx), columns the sweep skips (0, 0, 0); with phi = 0 outside the
// strips, C*curl + dt*(0*0 + 0*curl) is C*curl to the bit (the sign of a zero result aside).
template <int SIGN, typename T, int V>
__device__ __forceinline__ void term_x(Pack<T, V>& d, const Pack<T, V>& curl, const Lane3<T, V>& t, T dt, T* phi) {
    Pack<T, V> ph = zero_pack<T, V>();
    if (phi) ph = ldv<T, V>(phi);
#pragma unroll
    for (int e = 0; e < V; ++e) {
        const T p = t.xb.v[e] * ph.v[e] + t.xa.v[e] * curl.v[e];
        ph.v[e] = p;
        const T v = t.xC.v[e] * curl.v[e] + dt * p;
        d.v[e] = SIGN > 0 ? d.v[e] + v : d.v[e] - v;
    }
    if (phi) stv<T, V>(phi, ph);
}

// ---- where the E-half finds H and the H-half finds E --------------------------------------
// FULL storage: the arrays written by the previous half step.  LEAN storage: E and H are not
// resident; the consumer recomputes them from D/eps and B/mu with the same division the producer
// would have stored, so both modes yield identical bits.
template <typename T, int V, bool MU, bool LEAN>
__device__ __forceinline__ Pack<T, V> load_h(const K3<T>& a, int c, size_t idx) {
    if (!LEAN) return ldv<T, V>(a.H[c] + idx);
    const Pack<T, V> b = ldv<T, V>(a.B[c] + idx);
    return MU ? divide<T, V>(b, ldv<T, V>(a.mu[c] + idx)) : divide<T, V>(b, a.mu0);
}
template <typename T, bool MU, bool LEAN>
__device__ __forceinline__ T load_h1(const K3<T>& a, int c, size_t idx) {
    if (!LEAN) return a.H[c][idx];
    return MU ? a.B[c][idx] / a.mu[c][idx] : a.B[c][idx] / a.mu0;
}
template <typename T, int V, bool LEAN>
__device__ __forceinline__ Pack<T, V> load_e(const K3<T>& a, int c, size_t idx) {
    if (!LEAN) return ldv<T, V>(a.E[c] + idx);
    return divide<T, V>(ldv<T, V>(a.D[c] + idx), ldv<T, V>(a.eps[c] + idx));
}
template <typename T, bool LEAN>
__device__ __forceinline__ T load_e1(const K3<T>& a, int c, size_t idx) {
    if (!LEAN) return a.E[c][idx];
    return a.D[c][idx] / a.eps[c][idx];
}

// ------------------------------------------------------------------ E-half --
// D += curl H (CPML folded in), Dx += dt*wf[n]*mag on the source plane, E = D/eps.
template <typename T, bool MU, bool LEAN>
__global__ void __launch_bounds__(256) k3_half_e(const K3<T> a) {
    constexpr int V = VecWidth<T>::V;
    using P = Pack<T, V>;
    const Lane3<T, V> t = decode_lane<T, V, false>(a);
    if (!t.ok) return;                       // whole wave leaves: one row per wave
    const int j = t.j, i0 = t.i0, lane = t.lane;
    const size_t sz = (size_t)a.ny1 * a.px;
    const int Ny = a.Ny, Nz = a.Nz, px = a.px;

    int ly;
    const int zy = zone_e(j, Ny, a.np[1][0], a.np[1][1], ly);
    const T* face_y = a.tab + (size_t)(1 * 2) * 6 * kMaxLayers;
    const T* face_z = a.tab + (size_t)(2 * 2) * 6 * kMaxLayers;
    const bool row_lt = j < Ny, row_in = j >= 1 && j < Ny;

    size_t idx = ((size_t)t.klb * a.ny1 + j) * px + i0;
    P hx_km = zero_pack<T, V>(), hy_km = zero_pack<T, V>();
    if (t.klb + a.zoff >= 1) { hx_km = load_h<T, V, MU, LEAN>(a, 0, idx - sz); hy_km = load_h<T, V, MU, LEAN>(a, 1, idx - sz); }

    for (int kl = t.klb; kl < t.kle; ++kl, idx += sz) {
        const int k = kl + a.zoff;
        int lk;
        const int zk = zone_e(k, Nz, a.np[2][0], a.np[2][1], lk);
        const bool pl_lt = k < Nz, pl_in = k >= 1 && k < Nz;

        // every load of the plane is issued before the first use; all addresses are inside the
        // padded boxes (components that do not exist at this row/plane read zeros)
        const P hx = load_h<T, V, MU, LEAN>(a, 0, idx), hy = load_h<T, V, MU, LEAN>(a, 1, idx), hz = load_h<T, V, MU, LEAN>(a, 2, idx);
        P hx_jm = zero_pack<T, V>(), hz_jm = zero_pack<T, V>();
        if (j >= 1) { hx_jm = load_h<T, V, MU, LEAN>(a, 0, idx - px); hz_jm = load_h<T, V, MU, LEAN>(a, 2, idx - px); }
        P dx = ldv<T, V>(a.D[0] + idx), dy = ldv<T, V>(a.D[1] + idx), dz = ldv<T, V>(a.D[2] + idx);
        P ex = zero_pack<T, V>(), ey = ex, ez = ex;
        if (!LEAN) { ex = ldv<T, V>(a.eps[0] + idx); ey = ldv<T, V>(a.eps[1] + idx); ez = ldv<T, V>(a.eps[2] + idx); }

        T hy_l = lane_below(hy.v[V - 1]), hz_l = lane_below(hz.v[V - 1]);
        if (lane == 0) {
            hy_l = i0 > 0 ? load_h1<T, MU, LEAN>(a, 1, idx - 1) : (T)0;
            hz_l = i0 > 0 ? load_h1<T, MU, LEAN>(a, 2, idx - 1) : (T)0;
        }
        T* phx = t.in_n ? a.phE[1][0][0] + ((size_t)kl * a.ny1 + j) * a.xwn + i0
                        : (t.in_p ? a.phE[1][0][1] + ((size_t)kl * a.ny1 + j) * a.xwp + (i0 - a.xp0) : nullptr);
        T* phx2 = t.in_n ? a.phE[2][0][0] + ((size_t)kl * a.ny1 + j) * a.xwn + i0
                         : (t.in_p ? a.phE[2][0][1] + ((size_t)kl * a.ny1 + j) * a.xwp + (i0 - a.xp0) : nullptr);

        // ---- Dx: -dHy/dz (rows 0..Ny-1), then +dHz/dy (planes 1..Nz-1), then the source
        if (row_lt) term_u<-1, T, V>(dx, diff<T, V>(hy, hy_km), false, zk, lk, face_z, 0, a.dtd[2], a.dt,
                                     a.phE[0][2][0] + ((size_t)lk * a.ny1 + j) * px + i0, a.phE[0][2][1] + ((size_t)lk * a.ny1 + j) * px + i0);
        if (pl_in) term_u<+1, T, V>(dx, diff<T, V>(hz, hz_jm), false, zy, ly, face_y, 0, a.dtd[1], a.dt,
                                    a.phE[0][1][0] + ((size_t)kl * a.np[1][0] + ly) * px + i0, a.phE[0][1][1] + ((size_t)kl * a.np[1][1] + ly) * px + i0);
        if (a.src_on && k == a.kJ) {
            const P m = ldv<T, V>(a.mag + (size_t)j * px + i0);
#pragma unroll
            for (int e = 0; e < V; ++e) dx.v[e] = dx.v[e] + a.jx * m.v[e];
        }
        stv_nt<T, V>(a.D[0] + idx, dx);
        if (!LEAN) stv_nt<T, V>(a.E[0] + idx, divide<T, V>(dx, ex));
        // ---- Dy: +dHx/dz (columns 1..Nx-1), then -dHz/dx (planes 1..Nz-1)
        if (row_lt) {
            term_u<+1, T, V>(dy, diff<T, V>(hx, hx_km), t.first, zk, lk, face_z, 0, a.dtd[2], a.dt,
                             a.phE[1][2][0] + ((size_t)lk * a.ny1 + j) * px + i0, a.phE[1][2][1] + ((size_t)lk * a.ny1 + j) * px + i0);
            if (pl_in) term_x<-1, T, V>(dy, diff_back_x<T, V>(hz, hz_l), t, a.dt, phx);
            stv_nt<T, V>(a.D[1] + idx, dy);
            if (!LEAN) stv_nt<T, V>(a.E[1] + idx, divide<T, V>(dy, ey));
        }
        // ---- Dz: +dHy/dx (rows 1..Ny-1), then -dHx/dy (columns 1..Nx-1); planes 0..Nz-1
        if (pl_lt) {
            if (row_in) term_x<+1, T, V>(dz, diff_back_x<T, V>(hy, hy_l), t, a.dt, phx2);
            term_u<-1, T, V>(dz, diff<T, V>(hx, hx_jm), t.first, zy, ly, face_y, 0, a.dtd[1], a.dt,
                             a.phE[2][1][0] + ((size_t)kl * a.np[1][0] + ly) * px + i0, a.phE[2][1][1] + ((size_t)kl * a.np[1][1] + ly) * px + i0);
            stv_nt<T, V>(a.D[2] + idx, dz);
            if (!LEAN) stv_nt<T, V>(a.E[2] + idx, divide<T, V>(dz, ez));
        }
        hx_km = hx; hy_km = hy;
    }
}

// ------------------------------------------------------------------ H-half --
// B -= curl E (CPML folded in), H = B/mu0 (or B/mu).
template <typename T, bool MU, bool LEAN>
__global__ void __launch_bounds__(256) k3_half_h(const K3<T> a) {
    constexpr int V = VecWidth<T>::V;
    using P = Pack<T, V>;
    const Lane3<T, V> t = decode_lane<T, V, true>(a);
    if (!t.ok) return;
    const int j = t.j, i0 = t.i0, lane = t.lane;
    const size_t sz = (size_t)a.ny1 * a.px;
    const int Ny = a.Ny, Nz = a.Nz, px = a.px;

    int ly;
    const int zy = zone_h(j, Ny, a.np[1][0], a.np[1][1], true, ly);
    const T* face_y = a.tab + (size_t)(1 * 2) * 6 * kMaxLayers;
    const T* face_z = a.tab + (size_t)(2 * 2) * 6 * kMaxLayers;
    const bool row_lt = j < Ny, row_in = j >= 1 && j < Ny;

    size_t idx = ((size_t)t.klb * a.ny1 + j) * px + i0;
    P ex = load_e<T, V, LEAN>(a, 0, idx), ey = load_e<T, V, LEAN>(a, 1, idx);

    for (int kl = t.klb; kl < t.kle; ++kl, idx += sz) {
        const int k = kl + a.zoff;
        int lk;
        const int zk = zone_h(k, Nz, a.np[2][0], a.np[2][1], true, lk);
        const bool pl_lt = k < Nz, pl_in = k >= 1 && k < Nz;

        P ex_kp = zero_pack<T, V>(), ey_kp = zero_pack<T, V>();
        if (kl + 1 < a.nzp) { ex_kp = load_e<T, V, LEAN>(a, 0, idx + sz); ey_kp = load_e<T, V, LEAN>(a, 1, idx + sz); }
        const P ez = load_e<T, V, LEAN>(a, 2, idx);
        P ex_jp = zero_pack<T, V>(), ez_jp = zero_pack<T, V>();
        if (row_lt) { ex_jp = load_e<T, V, LEAN>(a, 0, idx + px); ez_jp = load_e<T, V, LEAN>(a, 2, idx + px); }
        P bx = ldv<T, V>(a.B[0] + idx), by = ldv<T, V>(a.B[1] + idx), bz = ldv<T, V>(a.B[2] + idx);
        P mx = zero_pack<T, V>(), my = mx, mz = mx;
        if (MU && !LEAN) { mx = ldv<T, V>(a.mu[0] + idx); my = ldv<T, V>(a.mu[1] + idx); mz = ldv<T, V>(a.mu[2] + idx); }

        T ey_r = lane_above(ey.v[0]), ez_r = lane_above(ez.v[0]);
        if (lane == kLanes - 1) {
            ey_r = i0 + V < px ? load_e1<T, LEAN>(a, 1, idx + V) : (T)0;
            ez_r = i0 + V < px ? load_e1<T, LEAN>(a, 2, idx + V) : (T)0;
        }
        T* phx1 = t.in_n ? a.phM[1][0][0] + ((size_t)kl * a.ny1 + j) * a.xwn + i0
                         : (t.in_p ? a.phM[1][0][1] + ((size_t)kl * a.ny1 + j) * a.xwp + (i0 - a.xp0) : nullptr);
        T* phx2 = t.in_n ? a.phM[2][0][0] + ((size_t)kl * a.ny1 + j) * a.xwn + i0
                         : (t.in_p ? a.phM[2][0][1] + ((size_t)kl * a.ny1 + j) * a.xwp + (i0 - a.xp0) : nullptr);

        // ---- Bx: +dEy/dz, then -dEz/dy; columns 1..Nx-1, rows 0..Ny-1, planes 0..Nz-1
        if (row_lt && pl_lt) {
            term_u<+1, T, V>(bx, diff<T, V>(ey_kp, ey), t.first, zk, lk, face_z, 3, a.dtd[2], a.dt,
                             a.phM[0][2][0] + ((size_t)lk * a.ny1 + j) * px + i0, a.phM[0][2][1] + ((size_t)lk * a.ny1 + j) * px + i0);
            term_u<-1, T, V>(bx, diff<T, V>(ez_jp, ez), t.first, zy, ly, face_y, 3, a.dtd[1], a.dt,
                             a.phM[0][1][0] + ((size_t)kl * a.np[1][0] + ly) * px + i0, a.phM[0][1][1] + ((size_t)kl * a.np[1][1] + ly) * px + i0);
            stv_nt<T, V>(a.B[0] + idx, bx);
            if (!LEAN) stv_nt<T, V>(a.H[0] + idx, MU ? divide<T, V>(bx, mx) : divide<T, V>(bx, a.mu0));
        }
        // ---- By: +dEz/dx, then -dEx/dz; rows 1..Ny-1 (row 0 and Ny only take H = B/mu), planes 0..Nz-1
        if (pl_lt) {
            if (row_in) {
                term_x<+1, T, V>(by, diff_fwd_x<T, V>(ez, ez_r), t, a.dt, phx1);
                term_u<-1, T, V>(by, diff<T, V>(ex_kp, ex), false, zk, lk, face_z, 3, a.dtd[2], a.dt,
                                 a.phM[1][2][0] + ((size_t)lk * a.ny1 + j) * px + i0, a.phM[1][2][1] + ((size_t)lk * a.ny1 + j) * px + i0);
            }
            stv_nt<T, V>(a.B[1] + idx, by);
            if (!LEAN) stv_nt<T, V>(a.H[1] + idx, MU ? divide<T, V>(by, my) : divide<T, V>(by, a.mu0));
        }
        // ---- Bz: +dEx/dy, then -dEy/dx; rows 0..Ny-1, planes 1..Nz-1 (planes 0 and Nz only take H = B/mu)
        if (row_lt) {
            if (pl_in) {
                term_u<+1, T, V>(bz, diff<T, V>(ex_jp, ex), false, zy, ly, face_y, 3, a.dtd[1], a.dt,
                                 a.phM[2][1][0] + ((size_t)kl * a.np[1][0] + ly) * px + i0, a.phM[2][1][1] + ((size_t)kl * a.np[1][1] + ly) * px + i0);
                term_x<-1, T, V>(bz, diff_fwd_x<T, V>(ey, ey_r), t, a.dt, phx2);
            }
            stv_nt<T, V>(a.B[2] + idx, bz);
            if (!LEAN) stv_nt<T, V>(a.H[2] + idx, MU ? divide<T, V>(bz, mz) : divide<T, V>(bz, a.mu0));
        }
        ex = ex_kp; ey = ey_kp;
    }
}

}  // namespace fdtd
