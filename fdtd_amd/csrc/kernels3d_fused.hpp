// kernels3d_fused.hpp -- one kernel per time step: E-half and H-half of the 3-D update fused.
//
// The two half-step kernels read and write D and B once each AND read the other family once more
// (the E-half reads B, the H-half reads D): 24 words per cell with eps and mu grids.  Here a
// workgroup marches through its planes doing, per plane k,
//     D_new(k) = D_old(k) + curl H_old      (H_old = B_old / mu, as the E-half)
//     B_new(k-1) = B_old(k-1) - curl E_new  (E_new = D_new / eps, as the H-half, one plane behind)
// so D, B, eps and mu cross HBM once per step: 18 words per cell plus halos.  What the lagging B
// update needs from the planes already visited travels on chip: E_new(k-1), B_old(k-1) and the
// two H_old(k-1) components stay in registers; the E_new row above (j+1) comes through LDS from the
// wave that owns that row; the column to the right (i+1) is one cross-lane shift.
//
// Halos are recomputed, never exchanged between workgroups: the last wave of a workgroup only
// produces E_new of the first row of the next tile, lane 63 recomputes Ey/Ez of the first column of
// the next x tile (a scalar side path), and a chunk of planes ends with one extra D iteration.
// Because a halo recomputation must see the *old* D, B and D-side CPML state of cells another
// workgroup may already have advanced, those arrays ping-pong between two buffers (read old, write
// new; the host swaps after the launch).  The B-side CPML state is only touched by its owner and is
// updated in place.  Per-cell arithmetic is the same sequence of operations as in the half-step
// kernels, so the results are bit-identical to them (and to the reference in fp64).
#pragma once
#include "kernels3d.hpp"

namespace fdtd {

// ---- addressing ------------------------------------------------------------------------------
// Everything but the lane's position inside its wave's row segment is wave-uniform (workgroup, row
// and plane).  Array bases are therefore formed on the scalar unit at the point of use and pinned
// into scalar registers, and every vector access is "scalar base + 32-bit lane offset" in the
// global address space.  The kernel argument block is re-read through a pointer pinned once per
// plane, so that the several dozen array pointers are fetched (scalar cache) where they are needed
// instead of living in registers across the loop.

template <typename T, int V>
__device__ __forceinline__ Pack<T, V> ldg(const T* ubase, unsigned lo) {
    typedef T vec __attribute__((ext_vector_type(V)));
    FDTD_AS(1) const T* g = (FDTD_AS(1) const T*)ubase;
    asm volatile("" : "+s"(g));
    const vec x = *(FDTD_AS(1) const vec*)(g + lo);
    Pack<T, V> r;
#pragma unroll
    for (int e = 0; e < V; ++e) r.v[e] = x[e];
    return r;
}
template <typename T, int V>
__device__ __forceinline__ void stg(T* ubase, unsigned lo, const Pack<T, V>& r) {
    typedef T vec __attribute__((ext_vector_type(V)));
    FDTD_AS(1) T* g = (FDTD_AS(1) T*)ubase;
    asm volatile("" : "+s"(g));
    vec x;
#pragma unroll
    for (int e = 0; e < V; ++e) x[e] = r.v[e];
    *(FDTD_AS(1) vec*)(g + lo) = x;
}
// Store of a stream nothing reads again during this step (the new D and B): non-temporal, so that it
// does not push the rows and planes neighbouring workgroups are about to reuse out of L2 (+1.3 %).
template <typename T, int V>
__device__ __forceinline__ void stg_nt(T* ubase, unsigned lo, const Pack<T, V>& r) {
    typedef T vec __attribute__((ext_vector_type(V)));
    FDTD_AS(1) T* g = (FDTD_AS(1) T*)ubase;
    asm volatile("" : "+s"(g));
    vec x;
#pragma unroll
    for (int e = 0; e < V; ++e) x[e] = r.v[e];
    __builtin_nontemporal_store(x, (FDTD_AS(1) vec*)(g + lo));
}
template <typename T>
__device__ __forceinline__ T ldg1(const T* uaddr) {
    FDTD_AS(1) const T* g = (FDTD_AS(1) const T*)uaddr;
    asm volatile("" : "+s"(g));
    return *g;
}
template <typename T> using ArgBlock = const FDTD_AS(4) K3<T>*;
template <typename T> using Slots = T* const FDTD_AS(4)*;      // two consecutive pointers: low side, high side

// Uniform-zone term with separate input/output CPML state and an optional store.  `in`/`out` point
// at the [low, high] pointer pair of the state arrays in the argument block; `off_n`/`off_p` are the
// uniform element offsets of the wave's segment in them.
template <int SIGN, typename T, int V>
__device__ __forceinline__ void term_u2(Pack<T, V>& d, const Pack<T, V>& curl, bool skip0, int zone, int l,
                                        const T* __restrict__ face, int kb, T cmid, T dt,
                                        Slots<T> in, Slots<T> out, size_t off_n, size_t off_p, unsigned lo, bool store) {
    if (zone == Z_NONE) return;
    const T d0 = d.v[0];
    if (zone == Z_MID) {
#pragma unroll
        for (int e = 0; e < V; ++e) { const T v = cmid * curl.v[e]; d.v[e] = SIGN > 0 ? d.v[e] + v : d.v[e] - v; }
    } else {
        const int side = zone == Z_POS;
        // (through the scalar cache: the layer is the same for the whole wave, and a vector load would make the term wait
        // for every load the wave has in flight)
        const FDTD_AS(4) T* t = (const FDTD_AS(4) T*)(face + side * 6 * kMaxLayers + kb * kMaxLayers + l);
        const T b = t[0], a = t[kMaxLayers], C = t[2 * kMaxLayers];
        const size_t off = side ? off_p : off_n;
        Pack<T, V> ph = ldg<T, V>(in[side] + off, lo);
        const T p0 = ph.v[0];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const T p = b * ph.v[e] + a * curl.v[e];
            ph.v[e] = p;
            const T v = C * curl.v[e] + dt * p;
            d.v[e] = SIGN > 0 ? d.v[e] + v : d.v[e] - v;
        }
        if (skip0) ph.v[0] = p0;
        if (store) stg<T, V>(out[side] + off, lo, ph);
    }
    if (skip0) d.v[0] = d0;
}
// The same for one scalar sample whose CPML state is read but never written (halo column).
template <int SIGN, typename T>
__device__ __forceinline__ void term_u1(T& d, T curl, int zone, int l, const T* __restrict__ face, T cmid, T dt,
                                        Slots<T> in, size_t off_n, size_t off_p) {
    if (zone == Z_NONE) return;
    T v;
    if (zone == Z_MID) v = cmid * curl;
    else {
        const int side = zone == Z_POS;
        const FDTD_AS(4) T* t = (const FDTD_AS(4) T*)(face + side * 6 * kMaxLayers + l);
        const T p = t[0] * ldg1<T>(in[side] + (side ? off_p : off_n)) + t[kMaxLayers] * curl;
        v = t[2 * kMaxLayers] * curl + dt * p;
    }
    d = SIGN > 0 ? d + v : d - v;
}
// x term with per-column C in registers; b and a are only fetched by lanes inside a CPML strip
// (`on_n`/`on_p`: this lane's pack lies in the low/high strip).  Every column carries its own
// coefficients: CPML columns the tabulated ones, plain columns (0, 0, dt/dx), skipped columns zeros.
template <int SIGN, typename T, int V>
__device__ __forceinline__ void term_x2(Pack<T, V>& d, const Pack<T, V>& curl, const Pack<T, V>& xC, const T* xr_b, int px,
                                        T dt, bool on_n, bool on_p, Slots<T> in, Slots<T> out, size_t off_n, ptrdiff_t off_p, unsigned lo, bool store) {
    if (on_n || on_p) {
        const Pack<T, V> b = ldg<T, V>(xr_b, lo), a = ldg<T, V>(xr_b + px, lo);
        Pack<T, V> ph = on_n ? ldg<T, V>(in[0] + off_n, lo) : ldg<T, V>(in[1] + off_p, lo);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const T p = b.v[e] * ph.v[e] + a.v[e] * curl.v[e];
            ph.v[e] = p;
            const T v = xC.v[e] * curl.v[e] + dt * p;
            d.v[e] = SIGN > 0 ? d.v[e] + v : d.v[e] - v;
        }
        if (store) { if (on_n) stg<T, V>(out[0] + off_n, lo, ph); else stg<T, V>(out[1] + off_p, lo, ph); }
    } else {
#pragma unroll
        for (int e = 0; e < V; ++e) { const T v = xC.v[e] * curl.v[e]; d.v[e] = SIGN > 0 ? d.v[e] + v : d.v[e] - v; }
    }
}
// The same with the CPML state `ph` already requested by the caller (so that its latency overlaps other work) and
// the per-column b, a read from LDS (`xl` = the tile's [b | a | C] rows of that field family, TW apart).
template <int SIGN, typename T, int V>
__device__ __forceinline__ void term_x3(Pack<T, V>& d, const Pack<T, V>& curl, const T* xl, int TW, Pack<T, V> ph,
                                        T dt, bool on_n, bool on_p, Slots<T> out, size_t off_n, ptrdiff_t off_p, unsigned lo, bool store) {
    const Pack<T, V> xC = ldv<T, V>(xl + 2 * TW + lo);
    if (on_n || on_p) {
        const Pack<T, V> b = ldv<T, V>(xl + lo), a = ldv<T, V>(xl + TW + lo);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const T p = b.v[e] * ph.v[e] + a.v[e] * curl.v[e];
            ph.v[e] = p;
            const T v = xC.v[e] * curl.v[e] + dt * p;
            d.v[e] = SIGN > 0 ? d.v[e] + v : d.v[e] - v;
        }
        if (store) { if (on_n) stg<T, V>(out[0] + off_n, lo, ph); else stg<T, V>(out[1] + off_p, lo, ph); }
    } else {
#pragma unroll
        for (int e = 0; e < V; ++e) { const T v = xC.v[e] * curl.v[e]; d.v[e] = SIGN > 0 ? d.v[e] + v : d.v[e] - v; }
    }
}
// H = B/mu of the wave's row segment at uniform element offset `uo`, and of one sample.
template <typename T, int V, bool MU>
__device__ __forceinline__ Pack<T, V> load_hu(ArgBlock<T> ap, int c, size_t uo, unsigned lo, T mu0) {
    const Pack<T, V> b = ldg<T, V>(ap->B[c] + uo, lo);
    return MU ? divide<T, V>(b, ldg<T, V>(ap->mu[c] + uo, lo)) : divide<T, V>(b, mu0);
}
template <typename T, bool MU>
__device__ __forceinline__ T load_hu1(ArgBlock<T> ap, int c, size_t uo, T mu0) {
    const T b = ldg1<T>(ap->B[c] + uo);
    return MU ? b / ldg1<T>(ap->mu[c] + uo) : b / mu0;
}

template <typename T, bool MU, int MAXW, int V>
__global__ void __launch_bounds__(kLanes * MAXW) __attribute__((amdgpu_waves_per_eu(4, 4))) k3_fused(const K3<T> a) {
    constexpr int TW = kLanes * V;                       // columns per wave
    using P = Pack<T, V>;
    extern __shared__ unsigned char smem_raw[];
    T* smem = reinterpret_cast<T*>(smem_raw);            // [2 planes][W rows][3 comps][TW]: E_new of the last two planes; [7][TW]: x coefficients
    ArgBlock<T> ap = (ArgBlock<T>)__builtin_amdgcn_kernarg_segment_ptr();

    const unsigned lane = threadIdx.x & (kLanes - 1);
    const unsigned lo = lane * V;                        // this lane's columns inside the wave's segment
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), W = a.wpb;
    unsigned bid = a.remap == 1 ? xcd_contiguous(blockIdx.x, gridDim.x) : (a.remap > 1 ? xcd_grouped(blockIdx.x, gridDim.x, (unsigned)a.remap) : blockIdx.x);
    const int bx = bid % a.nbx; bid /= a.nbx;
    const int by = bid % a.nby;
    const int bz = bid / a.nby;
    const int c0 = bx * TW;                              // first column of the wave's segment
    const int i0 = c0 + (int)lo;
    const int jr = by * (W - 1) + w;                     // the last wave is the halo row of the next tile
    const bool halo_row = w == W - 1;
    const bool row_ok = jr < a.ny1;
    const int j = row_ok ? jr : a.ny1 - 1;               // waves past the box keep valid addresses and join every barrier
    const int klb = a.kl0 + bz * a.chunk, kle = min(klb + a.chunk, a.kl1);
    const int kl_end = kle < a.kl_stop ? kle + 1 : kle;  // one extra D iteration closes the chunk
    const size_t sz = (size_t)a.ny1 * a.px;
    const int Nx = a.Nx, Ny = a.Ny, Nz = a.Nz, px = a.px;
    const bool first = i0 == 0;
    const bool in_n = a.np[0][0] > 0 && i0 < a.xwn;
    const bool in_p = a.np[0][1] > 0 && i0 >= a.xp0 && i0 < a.xp0 + a.xwp;
    // the x coefficients of the tile's columns -- b, a, C of the D side, then of the B side -- do not change from plane
    // to plane: they sit in LDS behind the E rows and are read where a term needs them, instead of occupying
    // registers through the loop or being fetched from memory in the middle of a plane
    T* const xs = smem + (size_t)2 * W * 3 * TW;
    for (int q = w; q < 6; q += W) stv<T, V>(xs + (size_t)q * TW + lo, ldg<T, V>(a.xrow + (size_t)q * px + c0, lo));
    const bool edge_wave = c0 + TW < Nx;                 // the next x tile exists
    const bool edge = lane == kLanes - 1 && edge_wave;   // this lane recomputes Ey, Ez of its first column
    T* const xe = xs + 6 * TW;                           // b, a, C (D side) of that column
    bool e_in_n = false, e_in_p = false;
    if (edge_wave) {
        const int ih = c0 + TW;
        if (w == 0 && lane < 3) xe[lane] = a.xrow[(size_t)lane * px + ih];
        e_in_n = a.np[0][0] > 0 && ih < a.xwn;
        e_in_p = a.np[0][1] > 0 && ih >= a.xp0 && ih < a.xp0 + a.xwp;
    }
    __syncthreads();
    // fp32: the single values a plane needs from outside the wave's row segment -- By, Bz (mu_y, mu_z) of the column
    // to the left for lane 0, and B, mu, Dy, Dz, eps_y, eps_z of the next tile's first column (and Bx, mu_x of the
    // row below it) for lane 63 -- are fetched by lanes 48..63, one value each, with ONE load per plane into ONE
    // register, and read from there with v_readlane where they are used: sixteen registers less in flight than one
    // load per value.  Each of those lanes keeps the base of "its" array and its offset from the row segment.
    constexpr bool EARLY = sizeof(T) == 4;
    const T* nb_base = a.B[0];
    ptrdiff_t nb_delta = 0;
    bool nb_on = false;
    if (EARLY) {
        const ptrdiff_t below = j >= 1 ? -(ptrdiff_t)px : 0;
        switch ((int)lane - 48) {
            case 0: nb_base = a.B[1]; nb_delta = -1; nb_on = c0 > 0; break;
            case 1: nb_base = a.B[2]; nb_delta = -1; nb_on = c0 > 0; break;
            case 2: nb_base = a.mu[1]; nb_delta = -1; nb_on = MU && c0 > 0; break;
            case 3: nb_base = a.mu[2]; nb_delta = -1; nb_on = MU && c0 > 0; break;
            case 4: nb_base = a.B[0]; nb_delta = TW; nb_on = edge_wave; break;
            case 5: nb_base = a.B[1]; nb_delta = TW; nb_on = edge_wave; break;
            case 6: nb_base = a.B[2]; nb_delta = TW; nb_on = edge_wave; break;
            case 7: nb_base = a.mu[0]; nb_delta = TW; nb_on = MU && edge_wave; break;
            case 8: nb_base = a.mu[1]; nb_delta = TW; nb_on = MU && edge_wave; break;
            case 9: nb_base = a.mu[2]; nb_delta = TW; nb_on = MU && edge_wave; break;
            case 10: nb_base = a.D[1]; nb_delta = TW; nb_on = edge_wave; break;
            case 11: nb_base = a.D[2]; nb_delta = TW; nb_on = edge_wave; break;
            case 12: nb_base = a.eps[1]; nb_delta = TW; nb_on = edge_wave; break;
            case 13: nb_base = a.eps[2]; nb_delta = TW; nb_on = edge_wave; break;
            case 14: nb_base = a.B[0]; nb_delta = TW + below; nb_on = edge_wave; break;
            case 15: nb_base = a.mu[0]; nb_delta = TW + below; nb_on = MU && edge_wave; break;
            default: break;
        }
    }

    int lye, lyh;
    const int zye = zone_e(j, Ny, a.np[1][0], a.np[1][1], lye);
    const int zyh = zone_h(j, Ny, a.np[1][0], a.np[1][1], true, lyh);
    const T* face_y = a.tab + (size_t)(1 * 2) * 6 * kMaxLayers;
    const T* face_z = a.tab + (size_t)(2 * 2) * 6 * kMaxLayers;
    const bool row_lt = j < Ny, row_in = j >= 1 && j < Ny;
    const bool mstore = a.peek == 0;      // (a launch whose results are only sampled leaves the in-place B-side CPML state alone)

    size_t ro = ((size_t)klb * a.ny1 + j) * px + c0;     // uniform offset of the wave's row segment in a field array
    P hx_km = zero_pack<T, V>(), hy_km = zero_pack<T, V>();
    T hxe_km = (T)0;
    if (klb + a.zoff >= 1) {
        hx_km = load_hu<T, V, MU>(ap, 0, ro - sz, lo, a.mu0); hy_km = load_hu<T, V, MU>(ap, 1, ro - sz, lo, a.mu0);
        if (edge) hxe_km = load_hu1<T, MU>(ap, 0, ro - sz + TW, a.mu0);
    }
    P bo_km[3] = {zero_pack<T, V>(), zero_pack<T, V>(), zero_pack<T, V>()};
    T eye_km = (T)0, eze_km = (T)0;

    for (int kl = klb; kl < kl_end; ++kl, ro += sz) {
        asm volatile("" : "+s"(ap));                     // array pointers are fetched again below, where they are used
        const int k = kl + a.zoff;
        const bool own_d = row_ok && !halo_row && kl < kle;
        int lke, lkh;
        const int zke = zone_e(k, Nz, a.np[2][0], a.np[2][1], lke);
        const int zkh = zone_h(k - 1, Nz, a.np[2][0], a.np[2][1], true, lkh);
        const bool pl_lt = k < Nz, pl_in = k >= 1 && k < Nz;

        // ------------------------------------------------------------ loads of plane k
        // EARLY (fp32): every vector load the plane needs outside the CPML zones is issued before the first arithmetic
        // on any of them, so that a wave waits for memory once per plane and not once per neighbour; loads whose
        // result is not needed (no row below, no column to the left) go to the own row or are skipped by their
        // lanes, never to a branch around a load-and-divide -- such a branch makes the wave wait for everything in
        // flight.  In fp64 the registers for that many loads in flight do not exist (the divisions need more
        // temporaries): there the row below and D, eps are requested after the own row's H has been formed.
        P bo[3];
        bo[0] = ldg<T, V>(ap->B[0] + ro, lo); bo[1] = ldg<T, V>(ap->B[1] + ro, lo); bo[2] = ldg<T, V>(ap->B[2] + ro, lo);
        P mu_[3];
        if (MU) { mu_[0] = ldg<T, V>(ap->mu[0] + ro, lo); mu_[1] = ldg<T, V>(ap->mu[1] + ro, lo); mu_[2] = ldg<T, V>(ap->mu[2] + ro, lo); }
        const size_t rj = j >= 1 ? ro - px : ro;            // the row below
        P b0j, b2j, m0j, m2j;
        auto request_row_below = [&]() {
            b0j = ldg<T, V>(ap->B[0] + rj, lo); b2j = ldg<T, V>(ap->B[2] + rj, lo);
            if (MU) { m0j = ldg<T, V>(ap->mu[0] + rj, lo); m2j = ldg<T, V>(ap->mu[2] + rj, lo); }
        };
        // the halo row only has to deliver Ex and Ez: its Dy and eps_y loads are pointed at the x components it
        // reads anyway (same cache lines, nothing new from memory); what it computes from them is never used
        const int cy = halo_row ? 0 : 1;
        P dx, dy, dz, epx, epy, epz;
        auto request_d_eps = [&]() {
            dx = ldg<T, V>(ap->D[0] + ro, lo); dy = ldg<T, V>(ap->D[cy] + ro, lo); dz = ldg<T, V>(ap->D[2] + ro, lo);
            epx = ldg<T, V>(ap->eps[0] + ro, lo); epy = ldg<T, V>(ap->eps[cy] + ro, lo); epz = ldg<T, V>(ap->eps[2] + ro, lo);
        };
        // lane 0: By, Bz (and their mu) of the column to the left of the tile
        T b1l = (T)0, b2l = (T)0, m1l = a.mu0, m2l = a.mu0;
        auto request_left = [&]() {
            if (lane == 0 && c0 > 0) {
                b1l = ldg1<T>(ap->B[1] + ro - 1); b2l = ldg1<T>(ap->B[2] + ro - 1);
                if (MU) { m1l = ldg1<T>(ap->mu[1] + ro - 1); m2l = ldg1<T>(ap->mu[2] + ro - 1); }
            }
        };
        T nb = (T)0;                                          // lanes 48..63: their neighbour value of this plane
        if (EARLY) {
            request_row_below(); request_d_eps();
            if (nb_on) nb = nb_base[(ptrdiff_t)ro + nb_delta];
        }
        auto nb_of = [&](int l) -> T {                        // lane l's value, as a wave-uniform scalar
            if constexpr (EARLY) return __builtin_bit_cast(T, __builtin_amdgcn_readlane(__builtin_bit_cast(int, nb), l));
            else return (T)0;
        };

        // ------------------------------------------------------------ H_old(k) = B_old / mu
        P hx, hy, hz;
        if (MU) { hx = divide<T, V>(bo[0], mu_[0]); hy = divide<T, V>(bo[1], mu_[1]); hz = divide<T, V>(bo[2], mu_[2]); }
        else { hx = divide<T, V>(bo[0], a.mu0); hy = divide<T, V>(bo[1], a.mu0); hz = divide<T, V>(bo[2], a.mu0); }
        P hx_jm = zero_pack<T, V>(), hz_jm = zero_pack<T, V>();
        if (j >= 1) {
            if (!EARLY) request_row_below();
            if (MU) { hx_jm = divide<T, V>(b0j, m0j); hz_jm = divide<T, V>(b2j, m2j); }
            else { hx_jm = divide<T, V>(b0j, a.mu0); hz_jm = divide<T, V>(b2j, a.mu0); }
        }
        if (!EARLY) { request_d_eps(); request_left(); }
        T hy_l = lane_below(hy.v[V - 1]), hz_l = lane_below(hz.v[V - 1]);
        if (EARLY && c0 > 0) { b1l = nb_of(48); b2l = nb_of(49); if (MU) { m1l = nb_of(50); m2l = nb_of(51); } }
        if (lane == 0) {
            hy_l = c0 > 0 ? b1l / m1l : (T)0;
            hz_l = c0 > 0 ? b2l / m2l : (T)0;
        }
        // lane 63: the first column of the next x tile (fp64: requested where it is used)
        T be0 = (T)0, be1 = (T)0, be2 = (T)0, me0 = a.mu0, me1 = a.mu0, me2 = a.mu0, d1e = (T)0, d2e = (T)0, ep1e = (T)1, ep2e = (T)1, be0j = (T)0, me0j = a.mu0;
        auto request_edge = [&]() {
            const size_t ie = ro + TW, iej = rj + TW;
            be0 = ldg1<T>(ap->B[0] + ie); be1 = ldg1<T>(ap->B[1] + ie); be2 = ldg1<T>(ap->B[2] + ie);
            if (MU) { me0 = ldg1<T>(ap->mu[0] + ie); me1 = ldg1<T>(ap->mu[1] + ie); me2 = ldg1<T>(ap->mu[2] + ie); }
            d1e = ldg1<T>(ap->D[1] + ie); d2e = ldg1<T>(ap->D[2] + ie);
            ep1e = ldg1<T>(ap->eps[1] + ie); ep2e = ldg1<T>(ap->eps[2] + ie);
            be0j = ldg1<T>(ap->B[0] + iej);
            if (MU) me0j = ldg1<T>(ap->mu[0] + iej);
        };
        if (EARLY && edge_wave) {
            be0 = nb_of(52); be1 = nb_of(53); be2 = nb_of(54);
            if (MU) { me0 = nb_of(55); me1 = nb_of(56); me2 = nb_of(57); }
            d1e = nb_of(58); d2e = nb_of(59); ep1e = nb_of(60); ep2e = nb_of(61);
            be0j = nb_of(62);
            if (MU) me0j = nb_of(63);
        }

        const size_t oz = ((size_t)lke * a.ny1 + j) * px + c0;                 // z-face state, D side (uniform)
        const size_t oy0 = ((size_t)kl * a.np[1][0] + lye) * px + c0, oy1 = ((size_t)kl * a.np[1][1] + lye) * px + c0;
        const size_t oxn = ((size_t)kl * a.ny1 + j) * a.xwn + c0;
        const ptrdiff_t oxp = (ptrdiff_t)(((size_t)kl * a.ny1 + j) * a.xwp) + (c0 - a.xp0);
        // lanes inside an x-face strip: the CPML state of their Dy and Dz terms, requested now and used further down
        P phy = zero_pack<T, V>(), phz = zero_pack<T, V>();
        if (in_n) { phy = ldg<T, V>(ap->phE[1][0][0] + oxn, lo); phz = ldg<T, V>(ap->phE[2][0][0] + oxn, lo); }
        else if (in_p) { phy = ldg<T, V>(ap->phE[1][0][1] + oxp, lo); phz = ldg<T, V>(ap->phE[2][0][1] + oxp, lo); }

        // ------------------------------------------------------------ D_new(k), E_new(k)
        if (row_lt) term_u2<-1, T, V>(dx, diff<T, V>(hy, hy_km), false, zke, lke, face_z, 0, a.dtd[2], a.dt, ap->phE[0][2], ap->phEn[0][2], oz, oz, lo, own_d);
        if (pl_in) term_u2<+1, T, V>(dx, diff<T, V>(hz, hz_jm), false, zye, lye, face_y, 0, a.dtd[1], a.dt, ap->phE[0][1], ap->phEn[0][1], oy0, oy1, lo, own_d);
        if (a.src_on && k == a.kJ) {
            const P m = ldg<T, V>(a.mag + (size_t)j * px + c0, lo);
#pragma unroll
            for (int e = 0; e < V; ++e) dx.v[e] = dx.v[e] + a.jx * m.v[e];
        }
        if (row_lt) {
            term_u2<+1, T, V>(dy, diff<T, V>(hx, hx_km), first, zke, lke, face_z, 0, a.dtd[2], a.dt, ap->phE[1][2], ap->phEn[1][2], oz, oz, lo, own_d);
            if (pl_in) term_x3<-1, T, V>(dy, diff_back_x<T, V>(hz, hz_l), xs, TW, phy, a.dt, in_n, in_p, ap->phEn[1][0], oxn, oxp, lo, own_d);
        }
        if (pl_lt) {
            if (row_in) term_x3<+1, T, V>(dz, diff_back_x<T, V>(hy, hy_l), xs, TW, phz, a.dt, in_n, in_p, ap->phEn[2][0], oxn, oxp, lo, own_d);
            term_u2<-1, T, V>(dz, diff<T, V>(hx, hx_jm), first, zye, lye, face_y, 0, a.dtd[1], a.dt, ap->phE[2][1], ap->phEn[2][1], oy0, oy1, lo, own_d);
        }
        if (own_d) {
            stg_nt<T, V>(ap->Dn[0] + ro, lo, dx);
            if (row_lt) stg_nt<T, V>(ap->Dn[1] + ro, lo, dy);
            if (pl_lt) stg_nt<T, V>(ap->Dn[2] + ro, lo, dz);
        }
        // the same for the By and Bz terms of plane k-1, in flight during the divisions below
        const bool do_b = kl > klb && row_ok && !halo_row;
        const size_t pxn = ((size_t)(kl - 1) * a.ny1 + j) * a.xwn + c0;
        const ptrdiff_t pxp = (ptrdiff_t)(((size_t)(kl - 1) * a.ny1 + j) * a.xwp) + (c0 - a.xp0);
        P pmy = zero_pack<T, V>(), pmz = zero_pack<T, V>();
        if (do_b) {
            if (in_n) { pmy = ldg<T, V>(ap->phM[1][0][0] + pxn, lo); pmz = ldg<T, V>(ap->phM[2][0][0] + pxn, lo); }
            else if (in_p) { pmy = ldg<T, V>(ap->phM[1][0][1] + pxp, lo); pmz = ldg<T, V>(ap->phM[2][0][1] + pxp, lo); }
        }
        P en[3];
        en[0] = divide<T, V>(dx, epx); en[1] = divide<T, V>(dy, epy); en[2] = divide<T, V>(dz, epz);

        // ------------------------------------------------------------ halo column of the next x tile (lane 63 only)
        T eye = (T)0, eze = (T)0, hxe = (T)0;
        if (edge) {
            if (!EARLY) request_edge();
            hxe = be0 / me0;
            const T hye = be1 / me1, hze = be2 / me2;
            T d1 = d1e, d2 = d2e;
            const ptrdiff_t oxe = e_in_n ? (ptrdiff_t)oxn + TW : oxp + TW;
            if (row_lt) {
                term_u1<+1, T>(d1, hxe - hxe_km, zke, lke, face_z, a.dtd[2], a.dt, ap->phE[1][2], oz + TW, oz + TW);
                if (pl_in) {
                    const T curl = hze - hz.v[V - 1];
                    const T ph = (e_in_n || e_in_p) ? ldg1<T>(ap->phE[1][0][e_in_p] + oxe) : (T)0;
                    const T p = xe[0] * ph + xe[1] * curl;
                    d1 = d1 - (xe[2] * curl + a.dt * p);
                }
            }
            if (pl_lt) {
                if (row_in) {
                    const T curl = hye - hy.v[V - 1];
                    const T ph = (e_in_n || e_in_p) ? ldg1<T>(ap->phE[2][0][e_in_p] + oxe) : (T)0;
                    const T p = xe[0] * ph + xe[1] * curl;
                    d2 = d2 + (xe[2] * curl + a.dt * p);
                }
                const T hxe_jm = j >= 1 ? be0j / me0j : (T)0;
                term_u1<-1, T>(d2, hxe - hxe_jm, zye, lye, face_y, a.dtd[1], a.dt, ap->phE[2][1], oy0 + TW, oy1 + TW);
            }
            eye = d1 / ep1e;
            eze = d2 / ep2e;
        }

        // ------------------------------------------------------------ B_new(k-1)
        if (do_b) {
            asm volatile("" : "+s"(ap));
            // the row above, plane k-1, published by the next wave at the end of the previous iteration
            const T* own = smem + ((size_t)(((kl - 1) & 1) * W + w) * 3) * TW + lo;     // this row, plane k-1
            const T* src = own + 3 * TW;                                                // the row above
            const P ex_jp = ldv<T, V>(src), ez_jp = ldv<T, V>(src + 2 * TW);
            P en_km[3];
            en_km[0] = ldv<T, V>(own); en_km[1] = ldv<T, V>(own + TW); en_km[2] = ldv<T, V>(own + 2 * TW);
            const size_t rb = ro - sz;
            const int klm = kl - 1, km = k - 1;
            const bool bl_lt = km < Nz, bl_in = km >= 1 && km < Nz;
            T ey_r = lane_above(en_km[1].v[0]), ez_r = lane_above(en_km[2].v[0]);
            if (lane == kLanes - 1) { ey_r = eye_km; ez_r = eze_km; }
            const size_t pz = ((size_t)lkh * a.ny1 + j) * px + c0;
            const size_t py0 = ((size_t)klm * a.np[1][0] + lyh) * px + c0, py1 = ((size_t)klm * a.np[1][1] + lyh) * px + c0;
            if (row_lt && bl_lt) {
                P b = bo_km[0];
                term_u2<+1, T, V>(b, diff<T, V>(en[1], en_km[1]), first, zkh, lkh, face_z, 3, a.dtd[2], a.dt, ap->phM[0][2], ap->phM[0][2], pz, pz, lo, mstore);
                term_u2<-1, T, V>(b, diff<T, V>(ez_jp, en_km[2]), first, zyh, lyh, face_y, 3, a.dtd[1], a.dt, ap->phM[0][1], ap->phM[0][1], py0, py1, lo, mstore);
                stg_nt<T, V>(ap->Bn[0] + rb, lo, b);
            }
            if (bl_lt) {
                P b = bo_km[1];
                if (row_in) {
                    term_x3<+1, T, V>(b, diff_fwd_x<T, V>(en_km[2], ez_r), xs + 3 * TW, TW, pmy, a.dt, in_n, in_p, ap->phM[1][0], pxn, pxp, lo, mstore);
                    term_u2<-1, T, V>(b, diff<T, V>(en[0], en_km[0]), false, zkh, lkh, face_z, 3, a.dtd[2], a.dt, ap->phM[1][2], ap->phM[1][2], pz, pz, lo, mstore);
                }
                stg_nt<T, V>(ap->Bn[1] + rb, lo, b);
            }
            if (row_lt) {
                P b = bo_km[2];
                if (bl_in) {
                    term_u2<+1, T, V>(b, diff<T, V>(ex_jp, en_km[0]), false, zyh, lyh, face_y, 3, a.dtd[1], a.dt, ap->phM[2][1], ap->phM[2][1], py0, py1, lo, mstore);
                    term_x3<-1, T, V>(b, diff_fwd_x<T, V>(en_km[1], ey_r), xs + 3 * TW, TW, pmz, a.dt, in_n, in_p, ap->phM[2][0], pxn, pxp, lo, mstore);
                }
                stg_nt<T, V>(ap->Bn[2] + rb, lo, b);
            }
        }

        // a slab below another one leaves B of its top layer to the kernel that runs once the upper
        // neighbour's D has arrived; the new buffer only receives the old values here -- from the launch
        // that owns the layer, not from a chunk below that merely revisits it as its closing plane (that
        // chunk may run after the layer has been finished)
        if (a.copy_top && kl == a.kl_stop - 1 && kl < kle && row_ok && !halo_row) {
            stg_nt<T, V>(ap->Bn[0] + ro, lo, bo[0]); stg_nt<T, V>(ap->Bn[1] + ro, lo, bo[1]); stg_nt<T, V>(ap->Bn[2] + ro, lo, bo[2]);
        }

        // ------------------------------------------------------------ hand-over to the next iteration
        {
            T* dst = smem + ((size_t)((kl & 1) * W + w) * 3) * TW + lo;
            stv<T, V>(dst, en[0]); stv<T, V>(dst + TW, en[1]); stv<T, V>(dst + 2 * TW, en[2]);
        }
        hx_km = hx; hy_km = hy; hxe_km = hxe;
        bo_km[0] = bo[0]; bo_km[1] = bo[1]; bo_km[2] = bo[2];
        eye_km = eye; eze_km = eze;
        __syncthreads();
    }
}

}  // namespace fdtd
