// kernels2d_fused.hpp -- the whole 2-D step (FDTD2D::stepOnce, fdtd2d/src/FDTD2D.cpp:275-482) in one kernel.
//
// The two-kernel path writes Ey, Hx, Hz to memory only for the other kernel to read them back: 12 words
// per cell and step even where no cell is dispersive.  Here a wave marches through a chunk of rows doing,
// per row k,
//     Dy_new(k) = Dy_old(k) + curl H_old,   Ey_new(k) = Lorentz(Dy_new/eps0)       (H_old = B_old/mu0)
//     Bx_new(k-1), Bz_new(k-1) = B_old(k-1) -/+ curl Ey_new,  H_new(k-1) = Lorentz(B_new/mu0)
// so that Dy, Bx, Bz cross memory once per step (6 words per cell) and E, H only exist in registers --
// except at cells with a non-zero Lorentz coefficient, whose samples and polarisation sums are kept in
// memory as the recursion needs them.  The rule is per cell: "C1, C2 or C3 non-zero here -> the stored
// sample is the field; otherwise the field is D/eps0 (B/mu0)", which is what the recursion yields
// there bit for bit; row segments without any such cell (flagged when the coefficients are uploaded)
// do not even load the coefficients.
//
// Nothing is exchanged between waves: x tiles overlap by one 16-byte pack (a wave computes Dy_new and
// Ey_new for 64 packs but owns -- stores, and updates B for -- the first 63), and a chunk of rows ends
// with one more D/E iteration whose results are not stored.  As a recomputation must see the old
// values, Dy, Bx, Bz, the D-side CPML state and the Ey/SEy samples are written to other buffers than
// they are read from (the host rotates them); the H samples and the B-side CPML state are only touched
// by their owner and are updated in place as in the two-kernel path.
#pragma once
#include "kernels2d.hpp"
#include "kernels3d_fused.hpp"

namespace fdtd {

template <typename T>
struct K2F {
    T* Dy; T* Bx; T* Bz;                           // state before the step (read only)
    T* Dyn; T* Bxn; T* Bzn;                        // state after the step
    T* Ec; T* Ep; T* En;                           // Ey samples n-1, n-2 and the buffer that receives the new one
    T* SEc; T* SEp; T* SEn;
    T* H1[2]; T* H2[2];                            // Hx, Hz: sample n-1 (read), n-2 (read, then overwritten with the new sample)
    T* SH1[2]; T* SH2[2];
    T* C1; T* C2; T* C3;
    T* pe_z[2]; T* pe_zn[2];                       // CPML state of Dy, z faces [low, high]: before / after
    T* pe_x[2]; T* pe_xn[2];                       // x faces
    T* pm_z[2]; T* pm_x[2];                        // CPML state of Bx (z faces), Bz (x faces): in place
    const T* tab;
    const T* xrow;                                 // [6][px]: per-column (b, a, C) of the x terms, plain columns (0, 0, dt/dx), skipped ones zeros
    int Nx, Nz, px, nz1;
    int np[2][2];
    int xwn, xwp, xp0;
    int iJ, kJ, src_on;
    T jy, dt, dtd[2], mu0, eps0;
    int chunk, nbx, nbz, ntx, remap;
    const unsigned char* disp;                     // [nz1][ntx]: any Lorentz coefficient non-zero in the 64 packs of that wave
};
template <typename T> using ArgBlock2 = const FDTD_AS(4) K2F<T>*;

// B/mu0 (or D/eps0) of a row segment, except at cells with a non-zero coefficient, where the stored sample is the field.
template <typename T, int V>
__device__ __forceinline__ Pack<T, V> field_of(const Pack<T, V>& raw, T scale, bool flagged, const T* stored, const T* c1, const T* c2, const T* c3, unsigned lo) {
    Pack<T, V> f = divide<T, V>(raw, scale);
    if (flagged) {
        const Pack<T, V> a = ldg<T, V>(c1, lo), b = ldg<T, V>(c2, lo), c = ldg<T, V>(c3, lo), s = ldg<T, V>(stored, lo);
#pragma unroll
        for (int e = 0; e < V; ++e) if (a.v[e] != (T)0 || b.v[e] != (T)0 || c.v[e] != (T)0) f.v[e] = s.v[e];
    }
    return f;
}
template <typename T>
__device__ __forceinline__ T field_of1(T raw, T scale, bool flagged, const T* stored, const T* c1, const T* c2, const T* c3) {
    T f = raw / scale;
    if (flagged && (ldg1<T>(c1) != (T)0 || ldg1<T>(c2) != (T)0 || ldg1<T>(c3) != (T)0)) f = ldg1<T>(stored);
    return f;
}

// Addressing as in the fused 3-D kernel: the row and the x tile are wave-uniform, so array bases are
// formed on the scalar unit where they are used, and the pointers are re-read from the argument block
// once per row instead of being kept in registers.  Columns past the grid hold zeros in every input,
// which keeps them zero without per-element masks; `first` leaves the wall column i = 0 alone where the
// reference's sweeps start at 1.
constexpr int kStack2D = 4;      // waves (= consecutive row chunks) per workgroup of the fused 2-D kernel

template <typename T>
__global__ void __launch_bounds__(kLanes * kStack2D) k2_fused(const K2F<T> a) {
    constexpr int V = VecWidth<T>::V;
    constexpr int OWN = (kLanes - 1) * V;                // columns a wave owns; it computes D and E for one pack more
    using P = Pack<T, V>;
    ArgBlock2<T> ap = (ArgBlock2<T>)__builtin_amdgcn_kernarg_segment_ptr();
    const int lane = threadIdx.x & (kLanes - 1);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned bid = a.remap == 1 ? xcd_contiguous(blockIdx.x, gridDim.x) : (a.remap > 1 ? xcd_grouped(blockIdx.x, gridDim.x, (unsigned)a.remap) : blockIdx.x);
    const int bx = bid % a.nbx, bz = bid / a.nbx;
    // the waves of a workgroup sit on consecutive row chunks of one x tile: the row that closes a wave's
    // chunk is the first row of the wave above, which hands its Ey_new over through LDS (one barrier per
    // kernel); only the top wave of a group computes its closing row itself
    __shared__ T handover[kStack2D][kLanes * V];
    const int tx = bx;
    const int c0 = tx * OWN, i0 = c0 + lane * V;
    const int px = a.px, Nz = a.Nz;
    const bool valid = i0 < px;                          // the overlap pack of the last tile may lie past the row
    const bool owned = valid && lane < kLanes - 1;
    const unsigned lo = valid ? lane * V : 0;            // lanes past the row re-read the tile's first pack and are zeroed
    const bool first = i0 == 0;
    const bool in_n = valid && a.np[0][0] > 0 && i0 < a.xwn;
    const bool in_p = valid && a.np[0][1] > 0 && i0 >= a.xp0 && i0 < a.xp0 + a.xwp;
    P xCe = ldg<T, V>(a.xrow + 2 * (size_t)px + c0, lo), xCm = ldg<T, V>(a.xrow + 5 * (size_t)px + c0, lo);
    if (!valid) { xCe = zero_pack<T, V>(); xCm = xCe; }
    const T* face_z = a.tab + (size_t)(1 * 2) * 6 * kMaxLayers;
    const int kb = (bz * kStack2D + w) * a.chunk, ke = min(kb + a.chunk, a.nz1);
    if (kb >= a.nz1) { __syncthreads(); return; }        // no rows left for this wave: it only joins the group's barrier
    const int k_end = ke < a.nz1 ? ke + 1 : ke;          // one more iteration closes the chunk
    const bool from_above = ke < a.nz1 && w + 1 < kStack2D;   // the closing row's Ey_new comes from the wave above
    const bool src_lane = valid && a.iJ >= i0 && a.iJ < i0 + V;

    // "any dispersive cell in this row segment / in the one to its left": fetched one row ahead, so that no row starts
    // by waiting for its flag
    unsigned char pf = a.disp[(size_t)kb * a.ntx + tx], pfl = c0 > 0 ? a.disp[(size_t)kb * a.ntx + tx - 1] : (unsigned char)0;

    size_t ro = (size_t)kb * px + c0;                    // uniform offset of the wave's row segment
    P hx_km = zero_pack<T, V>();
    if (kb >= 1) {
        const bool fl = __builtin_amdgcn_readfirstlane((int)a.disp[(size_t)(kb - 1) * a.ntx + tx]) != 0;
        const size_t q = ro - px;
        hx_km = field_of<T, V>(ldg<T, V>(ap->Bx + q, lo), a.mu0, fl, ap->H1[0] + q, ap->C1 + q, ap->C2 + q, ap->C3 + q, lo);
        if (!valid) hx_km = zero_pack<T, V>();
    }
    P e_km = zero_pack<T, V>(), bx_km = zero_pack<T, V>(), bz_km = zero_pack<T, V>();
    bool flag_km = false;

    for (int k = kb; k < k_end; ++k, ro += px) {
        asm volatile("" : "+s"(ap));
        const bool own_row = k < ke;
        const bool flag = __builtin_amdgcn_readfirstlane((int)pf) != 0, flag_left = __builtin_amdgcn_readfirstlane((int)pfl) != 0;
        if (k + 1 < k_end) {
            pf = a.disp[(size_t)(k + 1) * a.ntx + tx];
            if (c0 > 0) pfl = a.disp[(size_t)(k + 1) * a.ntx + tx - 1];
        }
        int lk;
        const int zk = zone_e(k, Nz, a.np[1][0], a.np[1][1], lk);
        const bool row_in = k >= 1 && k < Nz, row_lt = k < Nz;

        // ---------------------------------------------------------------- row k: H_old, Dy_new, Ey_new
        P ey = zero_pack<T, V>(), hx = ey, bxo = ey, bzo = ey;
        if (k == ke && from_above) {
            ey = ldv<T, V>(&handover[w + 1][lane * V]);
        } else {
        P d = ldg<T, V>(ap->Dy + ro, lo); bxo = ldg<T, V>(ap->Bx + ro, lo), bzo = ldg<T, V>(ap->Bz + ro, lo);
        T bz_left = (T)0;                                 // lane 0: Bz to the left of the tile, requested with the row
        if (lane == 0 && c0 > 0) bz_left = ldg1<T>(ap->Bz + ro - 1);
        if (!valid) { d = zero_pack<T, V>(); bxo = d; bzo = d; }
        hx = field_of<T, V>(bxo, a.mu0, flag, ap->H1[0] + ro, ap->C1 + ro, ap->C2 + ro, ap->C3 + ro, lo);
        P hz = field_of<T, V>(bzo, a.mu0, flag, ap->H1[1] + ro, ap->C1 + ro, ap->C2 + ro, ap->C3 + ro, lo);
        if (!valid) { hx = zero_pack<T, V>(); hz = hx; }
        T hz_l = lane_below(hz.v[V - 1]);
        if (lane == 0) {
            hz_l = (T)0;
            if (c0 > 0) {
                const size_t q = ro - 1;
                hz_l = field_of1<T>(bz_left, a.mu0, flag_left, ap->H1[1] + q, ap->C1 + q, ap->C2 + q, ap->C3 + q);
            }
        }
        const bool st = own_row && owned;
        const size_t oz = (size_t)lk * px + c0;
        const size_t oxn = (size_t)k * a.xwn + c0;
        const ptrdiff_t oxp = (ptrdiff_t)((size_t)k * a.xwp) + (c0 - a.xp0);
        term_u2<+1, T, V>(d, diff<T, V>(hx, hx_km), first, zk, lk, face_z, 0, a.dtd[1], a.dt, ap->pe_z, ap->pe_zn, oz, oz, lo, st);
        if (row_in) term_x2<-1, T, V>(d, diff_back_x<T, V>(hz, hz_l), xCe, a.xrow + c0, px, a.dt, in_n, in_p, ap->pe_x, ap->pe_xn, oxn, oxp, lo, st);
        if (a.src_on && k == a.kJ && src_lane) {
#pragma unroll
            for (int e = 0; e < V; ++e) if (i0 + e == a.iJ) d.v[e] = d.v[e] - a.jy;
        }
        if (st) stg_nt<T, V>(ap->Dyn + ro, lo, d);
        if (row_lt) {
            if (flag) {
                const P c1 = ldg<T, V>(ap->C1 + ro, lo), c2 = ldg<T, V>(ap->C2 + ro, lo), c3 = ldg<T, V>(ap->C3 + ro, lo);
                const P f1 = ldg<T, V>(ap->Ec + ro, lo), f2 = ldg<T, V>(ap->Ep + ro, lo), s1 = ldg<T, V>(ap->SEc + ro, lo), s2 = ldg<T, V>(ap->SEp + ro, lo);
                P sn;
#pragma unroll
                for (int e = 0; e < V; ++e) lorentz<T>(d.v[e] / a.eps0, c1.v[e], c2.v[e], c3.v[e], f1.v[e], f2.v[e], s1.v[e], s2.v[e], ey.v[e], sn.v[e]);
                if (st) { stg<T, V>(ap->En + ro, lo, ey); stg<T, V>(ap->SEn + ro, lo, sn); }
            } else {
                ey = divide<T, V>(d, a.eps0);
            }
            if (!valid) ey = zero_pack<T, V>();
        }
        }
        if (k == kb) {                                    // every wave with rows passes here exactly once
            stv<T, V>(&handover[w][lane * V], ey);
            __syncthreads();
        }

        // ---------------------------------------------------------------- row k-1: B_new, H_new
        const T e_r = lane_above(e_km.v[0]);          // taken with all lanes active: lane 62 needs lane 63's value
        if (k > kb) {
            asm volatile("" : "+s"(ap));
            const int km = k - 1;                        // km < Nz always: k <= Nz
            const size_t q = ro - px;
            int lm;
            const int zm = zone_h(km, Nz, a.np[1][0], a.np[1][1], false, lm);
            const size_t pz = (size_t)lm * px + c0;
            const size_t pxn = (size_t)km * a.xwn + c0;
            const ptrdiff_t pxp = (ptrdiff_t)((size_t)km * a.xwp) + (c0 - a.xp0);
            P c1 = zero_pack<T, V>(), c2 = c1, c3 = c1;
            if (flag_km) { c1 = ldg<T, V>(ap->C1 + q, lo); c2 = ldg<T, V>(ap->C2 + q, lo); c3 = ldg<T, V>(ap->C3 + q, lo); }
            {
                P b = bx_km;
                term_u2<+1, T, V>(b, diff<T, V>(ey, e_km), first, zm, lm, face_z, 3, a.dtd[1], a.dt, ap->pm_z, ap->pm_z, pz, pz, lo, owned);
                if (owned) stg_nt<T, V>(ap->Bxn + q, lo, b);
                if (flag_km) {
                    const P f1 = ldg<T, V>(ap->H1[0] + q, lo), s1 = ldg<T, V>(ap->SH1[0] + q, lo);
                    const P f2 = ldg<T, V>(ap->H2[0] + q, lo), s2 = ldg<T, V>(ap->SH2[0] + q, lo);
                    P fn, sn;
#pragma unroll
                    for (int e = 0; e < V; ++e) lorentz<T>(b.v[e] / a.mu0, c1.v[e], c2.v[e], c3.v[e], f1.v[e], f2.v[e], s1.v[e], s2.v[e], fn.v[e], sn.v[e]);
                    if (owned) { stg<T, V>(ap->H2[0] + q, lo, fn); stg<T, V>(ap->SH2[0] + q, lo, sn); }
                }
            }
            {
                P b = bz_km;
                if (km >= 1) term_x2<-1, T, V>(b, diff_fwd_x<T, V>(e_km, e_r), xCm, a.xrow + 3 * (size_t)px + c0, px, a.dt, in_n, in_p, ap->pm_x, ap->pm_x, pxn, pxp, lo, owned);
                if (owned) stg_nt<T, V>(ap->Bzn + q, lo, b);
                if (flag_km) {
                    const P f1 = ldg<T, V>(ap->H1[1] + q, lo), s1 = ldg<T, V>(ap->SH1[1] + q, lo);
                    const P f2 = ldg<T, V>(ap->H2[1] + q, lo), s2 = ldg<T, V>(ap->SH2[1] + q, lo);
                    P fn, sn;
#pragma unroll
                    for (int e = 0; e < V; ++e) lorentz<T>(b.v[e] / a.mu0, c1.v[e], c2.v[e], c3.v[e], f1.v[e], f2.v[e], s1.v[e], s2.v[e], fn.v[e], sn.v[e]);
                    if (owned) { stg<T, V>(ap->H2[1] + q, lo, fn); stg<T, V>(ap->SH2[1] + q, lo, sn); }
                }
            }
        }
        hx_km = hx; e_km = ey; bx_km = bxo; bz_km = bzo; flag_km = flag;
    }
}

// Flags of overlapping row segments: 64 packs starting every `own_packs` packs (63 for the one-step fused kernel, 62 for
// the two-step one).
template <typename T, int V>
__global__ void k2_flags_fused(const T* C1, const T* C2, const T* C3, unsigned char* disp, int px, int nz1, int ntx, int own_packs) {
    const int OWN = own_packs * V;
    const int lane = threadIdx.x & (kLanes - 1);
    const size_t tile = (size_t)blockIdx.x * (blockDim.x / kLanes) + (threadIdx.x >> 6);
    if (tile >= (size_t)nz1 * ntx) return;
    const int k = (int)(tile / ntx), tx = (int)(tile % ntx);
    const int i0 = tx * OWN + lane * V;
    bool any = false;
    if (i0 < px) {
        const size_t idx = (size_t)k * px + i0;
        const Pack<T, V> a = ldv<T, V>(C1 + idx), b = ldv<T, V>(C2 + idx), c = ldv<T, V>(C3 + idx);
#pragma unroll
        for (int e = 0; e < V; ++e) any = any || a.v[e] != (T)0 || b.v[e] != (T)0 || c.v[e] != (T)0;
    }
    const unsigned long long vote = __ballot(any);
    if (lane == 0) disp[tile] = vote != 0ull ? 1 : 0;
}

// Fields that only exist in registers during a step, formed for downloads, observables and probes:
// out = stored sample at cells with a non-zero coefficient (and in the rows the step never touches),
// raw / scale elsewhere.
template <typename T>
__global__ void k2_materialize(const T* raw, const T* stored, const T* C1, const T* C2, const T* C3, T* out, T scale, int px, int rows_all, int rows_derived) {
    const size_t total = (size_t)rows_all * px;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(t / px);
        const bool keep = k >= rows_derived || C1[t] != (T)0 || C2[t] != (T)0 || C3[t] != (T)0;
        out[t] = keep ? stored[t] : raw[t] / scale;
    }
}

}  // namespace fdtd
