// kernels3d_pair.hpp -- two time steps of the 3-D update in one pass over memory (no CPML).
//
// The one-step fused kernel moves D, B, eps and mu across HBM once per step.  Here a workgroup marches
// through its planes carrying TWO steps at once, so that they cross HBM once per two steps:
//     iteration at plane k:   A  D1(k)   = D0(k)   + curl H0        E1 = D1/eps     (step n,   as FDTD3D.cpp:480-806)
//                             B  B1(k-1) = B0(k-1) - curl E1        H1 = B1/mu      (step n,   FDTD3D.cpp:808-1021)
//                             C  D2(k-1) = D1(k-1) + curl H1        E2 = D2/eps     (step n+1)
//                             D  B2(k-2) = B1(k-2) - curl E2                        (step n+1)
// Only D2 and B2 are stored; D1, B1, E1, H1, E2 live in registers (own cells) and LDS (the rows that
// neighbouring waves need).  Per-cell arithmetic is the sequence of operations of the one-step kernels,
// so the results are bit-identical to two of their steps.
//
// Every tile recomputes what it needs from its neighbours (nothing is exchanged between workgroups):
//   x: a wave's 64 packs overlap the next tile's by 4 columns; the two outermost columns on either side
//      are computed but not stored (left: E1 has no H0 to its left; right: H1 has no E1 to its right);
//   y: a workgroup of W waves owns W-3 rows: one below (its H1 feeds the first owned row's D2), two above;
//   z: a chunk of C planes takes C+3 iterations (one plane below, two above).
// Old and new values of a cell are needed by different tiles at different times, so D and B ping-pong
// between two buffers exactly as in the one-step fused kernel.
//
// Restrictions (checked by the host): no CPML layers on any face, the whole grid on one device.
#pragma once
#include "kernels3d_fused.hpp"

namespace fdtd {

template <typename T, int V>
__device__ __forceinline__ Pack<T, V> lds_ld(const T* p) {
    typedef T vec __attribute__((ext_vector_type(V)));
    const vec x = *reinterpret_cast<const vec*>(p);
    Pack<T, V> r;
#pragma unroll
    for (int e = 0; e < V; ++e) r.v[e] = x[e];
    return r;
}
template <typename T, int V>
__device__ __forceinline__ void lds_st(T* p, const Pack<T, V>& r) {
    typedef T vec __attribute__((ext_vector_type(V)));
    vec x;
#pragma unroll
    for (int e = 0; e < V; ++e) x[e] = r.v[e];
    *reinterpret_cast<vec*>(p) = x;
}
template <typename T> __device__ __forceinline__ Pack<T, 1> lds_ld1(const T* p) { Pack<T, 1> r; r.v[0] = *p; return r; }

// d +- c * curl with a wave-uniform coefficient (the plain update between CPML layers); `skip0` leaves
// the first element alone (tangential component on the wall column i = 0)
template <int SIGN, typename T, int V>
__device__ __forceinline__ void pair_axpy(Pack<T, V>& d, T c, const Pack<T, V>& curl, bool skip0) {
    const T d0 = d.v[0];
#pragma unroll
    for (int e = 0; e < V; ++e) { const T v = c * curl.v[e]; d.v[e] = SIGN > 0 ? d.v[e] + v : d.v[e] - v; }
    if (skip0) d.v[0] = d0;
}
// the same with a coefficient per column (x terms: dt/dx, or 0 on the columns the sweeps leave out)
template <int SIGN, typename T, int V>
__device__ __forceinline__ void pair_axpy_x(Pack<T, V>& d, const Pack<T, V>& c, const Pack<T, V>& curl) {
#pragma unroll
    for (int e = 0; e < V; ++e) { const T v = c.v[e] * curl.v[e]; d.v[e] = SIGN > 0 ? d.v[e] + v : d.v[e] - v; }
}

// D of one plane from H of the same step: the terms, their order and their index ranges are those of k3_fused
// with every zone "plain" (1 <= k < Nz for the z term and so on).
template <typename T, int V>
__device__ __forceinline__ void pair_update_d(Pack<T, V>& dx, Pack<T, V>& dy, Pack<T, V>& dz,
                                              const Pack<T, V>& hx, const Pack<T, V>& hy, const Pack<T, V>& hz,
                                              const Pack<T, V>& hx_km, const Pack<T, V>& hy_km,
                                              const Pack<T, V>& hx_jm, const Pack<T, V>& hz_jm, T hy_l, T hz_l,
                                              const Pack<T, V>& xc, T cy, T cz,
                                              bool row_lt, bool row_in, bool pl_lt, bool pl_in, bool first) {
    if (row_lt && pl_in) pair_axpy<-1, T, V>(dx, cz, diff<T, V>(hy, hy_km), false);
    if (pl_in && row_in) pair_axpy<+1, T, V>(dx, cy, diff<T, V>(hz, hz_jm), false);
    if (row_lt) {
        if (pl_in) {
            pair_axpy<+1, T, V>(dy, cz, diff<T, V>(hx, hx_km), first);
            pair_axpy_x<-1, T, V>(dy, xc, diff_back_x<T, V>(hz, hz_l));
        }
    }
    if (pl_lt) {
        if (row_in) {
            pair_axpy_x<+1, T, V>(dz, xc, diff_back_x<T, V>(hy, hy_l));
            pair_axpy<-1, T, V>(dz, cy, diff<T, V>(hx, hx_jm), first);
        }
    }
}
// B of plane km from E of the same step at planes km (`e`) and km+1 (`e_kp`), the row above (`ex_jp`, `ez_jp`) and
// the column to the right (`ey_r`, `ez_r`).
template <typename T, int V>
__device__ __forceinline__ void pair_update_b(Pack<T, V>& bx, Pack<T, V>& by, Pack<T, V>& bz,
                                              const Pack<T, V>* e, const Pack<T, V>* e_kp,
                                              const Pack<T, V>& ex_jp, const Pack<T, V>& ez_jp, T ey_r, T ez_r,
                                              const Pack<T, V>& xc, T cy, T cz,
                                              bool row_lt, bool row_in, bool bl_lt, bool bl_in, bool zk_mid, bool zy_mid, bool first) {
    if (row_lt && bl_lt) {
        if (zk_mid) pair_axpy<+1, T, V>(bx, cz, diff<T, V>(e_kp[1], e[1]), first);
        if (zy_mid) pair_axpy<-1, T, V>(bx, cy, diff<T, V>(ez_jp, e[2]), first);
    }
    if (bl_lt && row_in) {
        pair_axpy_x<+1, T, V>(by, xc, diff_fwd_x<T, V>(e[2], ez_r));
        if (zk_mid) pair_axpy<-1, T, V>(by, cz, diff<T, V>(e_kp[0], e[0]), false);
    }
    if (row_lt && bl_in) {
        if (zy_mid) pair_axpy<+1, T, V>(bz, cy, diff<T, V>(ex_jp, e[0]), false);
        pair_axpy_x<-1, T, V>(bz, xc, diff_fwd_x<T, V>(e[1], ey_r));
    }
}

template <typename T, bool MU, int V>
__global__ void __launch_bounds__(kLanes * 16) __attribute__((amdgpu_waves_per_eu(4, 4))) k3_pair(const K3<T> a) {
    static_assert(V <= 2, "the two halo columns on either side must be whole packs");
    constexpr int TW = kLanes * V;                       // columns a wave computes
    constexpr int XS = TW - 4;                           // columns it owns
    using P = Pack<T, V>;
    extern __shared__ unsigned char smem_raw[];
    T* smem = reinterpret_cast<T*>(smem_raw);
    ArgBlock<T> ap = (ArgBlock<T>)__builtin_amdgcn_kernarg_segment_ptr();

    const unsigned lane = threadIdx.x & (kLanes - 1);
    const unsigned lo = lane * V;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), W = a.wpb;
    unsigned bid = a.remap > 1 ? xcd_grouped(blockIdx.x, gridDim.x, (unsigned)a.remap) : blockIdx.x;
    const int bx = bid % a.nbx; bid /= a.nbx;
    const int by = bid % a.nby;
    const int bz = bid / a.nby;
    const int px = a.px, Nx = a.Nx, Ny = a.Ny, Nz = a.Nz;
    (void)Nx;
    const int c0 = bx * XS;
    const int i0 = c0 + (int)lo;
    const bool live = i0 < px;
    // lanes past the end of the row (last x tile only) read the row's last pack instead: whatever they compute is
    // neither stored nor seen by a live lane (the column right of the last stored one is the wall: zero, below)
    const unsigned lc = live ? lo : (unsigned)(px - V - c0);
    const bool own_x = live && (bx == 0 || i0 >= c0 + 2) && i0 + V <= c0 + TW - 2;
    const bool wall_r = i0 + V >= px;
    const bool first = i0 == 0;
    const int jr = by * (W - 3) - 1 + w;
    const bool row_ok = jr >= 0 && jr < a.ny1;
    const int j = jr < 0 ? 0 : (jr < a.ny1 ? jr : a.ny1 - 1);
    const bool own = own_x && row_ok && w >= 1 && w < W - 2;
    const int kb = a.kl0 + bz * a.chunk, ke = min(kb + a.chunk, a.kl1);
    const size_t sz = (size_t)a.ny1 * px;
    const bool row_lt = j < Ny, row_in = j >= 1 && j < Ny, zy_mid_h = j <= Ny - 2;
    const T cy = a.dtd[1], cz = a.dtd[2];

    // LDS: [2][TW] x coefficients (D side, B side); then three regions [2 planes][W rows][2 comps][TW]:
    // E1 (x, z) for the row below, H1 (x, z) for the row above, E2 (x, z) for the row below
    T* const xs = smem;
    const size_t region = (size_t)2 * W * 2 * TW;
    T* const e1s = smem + 2 * TW;
    T* const h1s = e1s + region;
    T* const e2s = h1s + region;
    if (w < 2) lds_st<T, V>(xs + (size_t)w * TW + lo, ldg<T, V>(a.xrow + (size_t)(w == 0 ? 2 : 5) * px + c0, lc));
    __syncthreads();
    const P xcd = lds_ld<T, V>(xs + lo), xch = lds_ld<T, V>(xs + TW + lo);
    auto slot = [&](T* reg, int par, int row) -> T* { return reg + ((size_t)(par * W + row) * 2) * TW + lo; };
    const int w_up = w + 1 < W ? w + 1 : w, w_dn = w >= 1 ? w - 1 : w;

    auto h_of = [&](const P& b, const P& m) -> P { return MU ? divide<T, V>(b, m) : divide<T, V>(b, a.mu0); };
    auto left_of_tile = [&](T shifted) -> T { return (lane == 0 && bx == 0) ? (T)0 : shifted; };
    auto right_of_tile = [&](T shifted) -> T { return wall_r ? (T)0 : shifted; };

    // H0 of the plane below the first one visited
    P hx_km = zero_pack<T, V>(), hy_km = zero_pack<T, V>();
    if (kb - 2 >= 0 && kb - 2 + a.zoff >= 0) {
        const size_t r2 = ((size_t)(kb - 2) * a.ny1 + j) * px + c0;
        P m0 = zero_pack<T, V>(), m1 = zero_pack<T, V>();
        if (MU) { m0 = ldg<T, V>(ap->mu[0] + r2, lc); m1 = ldg<T, V>(ap->mu[1] + r2, lc); }
        hx_km = h_of(ldg<T, V>(ap->B[0] + r2, lc), m0); hy_km = h_of(ldg<T, V>(ap->B[1] + r2, lc), m1);
    }
    P b0_km[3], mu_km[3], e1_km[3], d1_km[3], ep_km[3];        // step-n state of plane k-1
    P h1_km2[2], b1_km2[3], e2_km2[3];                         // step-n+1 inputs of plane k-2
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        b0_km[c] = mu_km[c] = e1_km[c] = d1_km[c] = ep_km[c] = b1_km2[c] = e2_km2[c] = zero_pack<T, V>();
    }
    h1_km2[0] = h1_km2[1] = zero_pack<T, V>();

    for (int kl = kb - 1; kl <= ke + 1; ++kl) {
        asm volatile("" : "+s"(ap));
        const int klc = kl < a.nzp ? kl : a.nzp - 1;          // iterations past the last plane only drain the pipeline
        const size_t ro = ((size_t)klc * a.ny1 + j) * px + c0;
        const int k = kl + a.zoff, km = k - 1, km2 = k - 2;
        const int par = kl & 1;

        // ------------------------------------------------------------ A: D1(k), E1(k)
        P bo[3], mu_[3], d1[3], ep[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) bo[c] = ldg<T, V>(ap->B[c] + ro, lc);
        if (MU) {
#pragma unroll
            for (int c = 0; c < 3; ++c) mu_[c] = ldg<T, V>(ap->mu[c] + ro, lc);
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) mu_[c] = zero_pack<T, V>();
        }
        const size_t rj = j >= 1 ? ro - px : ro;
        const P b0j = ldg<T, V>(ap->B[0] + rj, lc), b2j = ldg<T, V>(ap->B[2] + rj, lc);
        P m0j = zero_pack<T, V>(), m2j = zero_pack<T, V>();
        if (MU) { m0j = ldg<T, V>(ap->mu[0] + rj, lc); m2j = ldg<T, V>(ap->mu[2] + rj, lc); }
#pragma unroll
        for (int c = 0; c < 3; ++c) { d1[c] = ldg<T, V>(ap->D[c] + ro, lc); ep[c] = ldg<T, V>(ap->eps[c] + ro, lc); }

        const P hx = h_of(bo[0], mu_[0]), hy = h_of(bo[1], mu_[1]), hz = h_of(bo[2], mu_[2]);
        P hx_jm = zero_pack<T, V>(), hz_jm = zero_pack<T, V>();
        if (j >= 1) { hx_jm = h_of(b0j, m0j); hz_jm = h_of(b2j, m2j); }
        {
            const T hy_l = left_of_tile(__shfl_up(hy.v[V - 1], 1)), hz_l = left_of_tile(__shfl_up(hz.v[V - 1], 1));
            const bool pl_lt = k >= 0 && k < Nz, pl_in = k >= 1 && k < Nz;
            pair_update_d<T, V>(d1[0], d1[1], d1[2], hx, hy, hz, hx_km, hy_km, hx_jm, hz_jm, hy_l, hz_l, xcd, cy, cz,
                                row_lt, row_in, pl_lt, pl_in, first);
            if ((a.src_on & 1) && k == a.kJ) {
                const P m = ldg<T, V>(a.mag + (size_t)j * px + c0, lc);
#pragma unroll
                for (int e = 0; e < V; ++e) d1[0].v[e] = d1[0].v[e] + a.jx * m.v[e];
            }
        }
        P e1[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) e1[c] = divide<T, V>(d1[c], ep[c]);
        lds_st<T, V>(slot(e1s, par, w), e1[0]); lds_st<T, V>(slot(e1s, par, w) + TW, e1[2]);

        // ------------------------------------------------------------ B: B1(k-1), H1(k-1)
        P b1[3], h1[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { b1[c] = b0_km[c]; h1[c] = zero_pack<T, V>(); }
        if (kl >= kb) {
            const T* up = slot(e1s, par ^ 1, w_up);              // E1(k-1) of the row above, published one iteration ago
            const P ex_jp = lds_ld<T, V>(up), ez_jp = lds_ld<T, V>(up + TW);
            const T ey_r = right_of_tile(__shfl_down(e1_km[1].v[0], 1)), ez_r = right_of_tile(__shfl_down(e1_km[2].v[0], 1));
            const bool bl_lt = km >= 0 && km < Nz, bl_in = km >= 1 && km < Nz, zk_mid = km >= 0 && km <= Nz - 2;
            pair_update_b<T, V>(b1[0], b1[1], b1[2], e1_km, e1, ex_jp, ez_jp, ey_r, ez_r, xch, cy, cz,
                                row_lt, row_in, bl_lt, bl_in, zk_mid, zy_mid_h, first);
#pragma unroll
            for (int c = 0; c < 3; ++c) h1[c] = h_of(b1[c], mu_km[c]);
            lds_st<T, V>(slot(h1s, par, w), h1[0]); lds_st<T, V>(slot(h1s, par, w) + TW, h1[2]);
        }
        __syncthreads();

        // ------------------------------------------------------------ C: D2(k-1), E2(k-1)
        P e2[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) e2[c] = zero_pack<T, V>();
        if (kl >= kb + 1) {
            asm volatile("" : "+s"(ap));
            const T* dn = slot(h1s, par, w_dn);                  // H1(k-1) of the row below, published before the barrier
            P hx_jm1 = lds_ld<T, V>(dn), hz_jm1 = lds_ld<T, V>(dn + TW);
            if (j < 1) { hx_jm1 = zero_pack<T, V>(); hz_jm1 = zero_pack<T, V>(); }
            const T hy_l = left_of_tile(__shfl_up(h1[1].v[V - 1], 1)), hz_l = left_of_tile(__shfl_up(h1[2].v[V - 1], 1));
            const bool pl_lt = km >= 0 && km < Nz, pl_in = km >= 1 && km < Nz;
            P d2[3] = {d1_km[0], d1_km[1], d1_km[2]};
            pair_update_d<T, V>(d2[0], d2[1], d2[2], h1[0], h1[1], h1[2], h1_km2[0], h1_km2[1], hx_jm1, hz_jm1, hy_l, hz_l, xcd, cy, cz,
                                row_lt, row_in, pl_lt, pl_in, first);
            if ((a.src_on & 2) && km == a.kJ) {
                const P m = ldg<T, V>(a.mag + (size_t)j * px + c0, lc);
#pragma unroll
                for (int e = 0; e < V; ++e) d2[0].v[e] = d2[0].v[e] + a.jx2 * m.v[e];
            }
            if (own && kl - 1 < ke) {
                const size_t rs = ro - sz * (size_t)(klc - (kl - 1));
                stg_nt<T, V>(ap->Dn[0] + rs, lo, d2[0]);
                if (row_lt) stg_nt<T, V>(ap->Dn[1] + rs, lo, d2[1]);
                if (pl_lt) stg_nt<T, V>(ap->Dn[2] + rs, lo, d2[2]);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) e2[c] = divide<T, V>(d2[c], ep_km[c]);
            lds_st<T, V>(slot(e2s, par, w), e2[0]); lds_st<T, V>(slot(e2s, par, w) + TW, e2[2]);
        }

        // ------------------------------------------------------------ D: B2(k-2)
        if (kl >= kb + 2) {
            const T* up = slot(e2s, par ^ 1, w_up);              // E2(k-2) of the row above, published one iteration ago
            const P ex_jp = lds_ld<T, V>(up), ez_jp = lds_ld<T, V>(up + TW);
            const T ey_r = right_of_tile(__shfl_down(e2_km2[1].v[0], 1)), ez_r = right_of_tile(__shfl_down(e2_km2[2].v[0], 1));
            const bool bl_lt = km2 >= 0 && km2 < Nz, bl_in = km2 >= 1 && km2 < Nz, zk_mid = km2 >= 0 && km2 <= Nz - 2;
            P b2[3] = {b1_km2[0], b1_km2[1], b1_km2[2]};
            pair_update_b<T, V>(b2[0], b2[1], b2[2], e2_km2, e2, ex_jp, ez_jp, ey_r, ez_r, xch, cy, cz,
                                row_lt, row_in, bl_lt, bl_in, zk_mid, zy_mid_h, first);
            if (own) {
                const size_t rs = ro - sz * (size_t)(klc - (kl - 2));
                if (row_lt && bl_lt) stg_nt<T, V>(ap->Bn[0] + rs, lo, b2[0]);
                if (bl_lt) stg_nt<T, V>(ap->Bn[1] + rs, lo, b2[1]);
                if (row_lt) stg_nt<T, V>(ap->Bn[2] + rs, lo, b2[2]);
            }
        }

        // ------------------------------------------------------------ hand-over to the next iteration
        hx_km = hx; hy_km = hy;
        h1_km2[0] = h1[0]; h1_km2[1] = h1[1];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            b0_km[c] = bo[c]; mu_km[c] = mu_[c]; e1_km[c] = e1[c]; d1_km[c] = d1[c]; ep_km[c] = ep[c];
            b1_km2[c] = b1[c]; e2_km2[c] = e2[c];
        }
    }
}

}  // namespace fdtd
