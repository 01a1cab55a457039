// kernels2d_pair.hpp -- two time steps of the 2-D update (FDTD2D::stepOnce twice, fdtd2d/src/FDTD2D.cpp:275-482) in one
// pass over memory.
//
// The one-step fused kernel moves Dy, Bx, Bz across HBM once per step.  Here a wave marches through a chunk of rows
// carrying TWO steps at once, so that they cross HBM once per two steps:
//     iteration at row k:   A  Dy1(k)   = Dy0(k)   + curl H0        Ey1 = Lorentz(Dy1/eps0)      (step n,   FDTD2D.cpp:289-379)
//                           B  B1(k-1)  = B0(k-1)  -/+ curl Ey1     H1  = Lorentz(B1/mu0)        (step n,   FDTD2D.cpp:383-481)
//                           C  Dy2(k-1) = Dy1(k-1) + curl H1        Ey2 = Lorentz(Dy2/eps0)      (step n+1)
//                           D  B2(k-2)  = B1(k-2)  -/+ curl Ey2     H2  = Lorentz(B2/mu0)        (step n+1)
// Only Dy2 and B2 are stored (and, at rows segments with Lorentz cells, the samples and polarisation sums of both steps);
// everything between lives in the wave's registers.  Per-cell arithmetic is the sequence of operations of the one-step
// kernel, so the results are bit-identical to two of its steps.
//
// Nothing is exchanged between waves: an x tile's 64 lanes (8 bytes each: two fp32 or one fp64 column) overlap the next
// tile's by two columns either side, which is what two steps reach (computed, not stored), and a chunk of C rows takes
// C + 3 iterations.  Eight bytes per lane instead of the one-step kernel's sixteen keep the 17 packs a wave hands from
// iteration to iteration (and with them the whole kernel) within 128 registers; the rows of the next iteration are
// requested before the current one is computed.  A halo cell recomputes its first step from the OLD state after its owner may have stored the new one,
// so everything the launch writes goes to other buffers than it reads: Dy, Bx, Bz, the CPML state of both sides, and
// four buffers per sample / sum family (n-1 and n-2 in; after the first and after the second step out).  The CPML state
// and the polarisation sum of a cell wait in registers between its two steps.
#pragma once
#include <type_traits>
#include "kernels2d_fused.hpp"

namespace fdtd {

template <typename T>
struct K2P {
    T* Dy; T* Bx; T* Bz;                           // state before the two steps (read only)
    T* Dyn; T* Bxn; T* Bzn;                        // state after them
    T* F1[3]; T* F2[3];                            // samples n-1, n-2 of Ey, Hx, Hz (read only)
    T* FX[3]; T* FY[3];                            // samples after the first and after the second step (written)
    T* S1[3]; T* S2[3]; T* SX[3]; T* SY[3];        // the same for the polarisation sums
    T* C1; T* C2; T* C3;
    T* pe_z[2]; T* pe_zn[2];                       // CPML state of Dy, z faces [low, high]: before / after
    T* pe_x[2]; T* pe_xn[2];                       // x faces
    T* pm_z[2]; T* pm_zn[2];                       // CPML state of Bx (z faces)
    T* pm_x[2]; T* pm_xn[2];                       // of Bz (x faces)
    const T* tab;
    const T* xrow;                                 // [6][px]: per-column (b, a, C), as for the one-step kernel
    int Nx, Nz, px, nz1;
    int np[2][2];
    int xwn, xwp, xp0;
    int iJ, kJ, src_on;                            // src_on: bit 0 = the first step has a source sample, bit 1 = the second
    T jy, jy2, dt, dtd[2], mu0, eps0;
    ConstDiv<T> dm, de;                            // division by mu0 and by eps0 (see ConstDiv)
    int chunk, ntx, nbx, nbz, remap;
    const unsigned char* disp;                     // [nz1][ntx]: any Lorentz coefficient non-zero in the 64 packs of that wave
    const int* tiles; int n_tiles;                 // the x tiles this launch covers
    // ACC: the running RMS window of Ey rides along (FDTD2D::run, FDTD2D.cpp:515-522: acc(i,k) (+)= Ey^2 * weight after
    // every step); flags: 1 = the first step opens a new window, 2 = the second does, 4 = keep the window as the first
    // step leaves it in `acc_snap`
    T* acc; T* acc_snap; T acc_w; int acc_flags;
};
template <typename T> using ArgBlock2P = const FDTD_AS(4) K2P<T>*;

constexpr int kPairTiles2D = 4;      // waves (= neighbouring x tiles of one chunk of rows) per workgroup

// lanes at either end of a wave that are computed but not stored: two columns
template <int V> constexpr int pair2d_halo_lanes() { return V >= 2 ? 1 : 2; }

// LZ = false: for x tiles none of whose row segments holds a Lorentz cell -- the kernel without the recursion, its
// loads and its registers (more waves per SIMD); the tiles that do hold such cells go through LZ = true in a launch of their own.
template <typename T, int V, bool LZ, bool ACC = false>
__global__ void __launch_bounds__(kLanes * kPairTiles2D) k2_pair(const K2P<T> a) {
    constexpr int HL = pair2d_halo_lanes<V>();
    constexpr int OWN = (kLanes - 2 * HL) * V;           // columns a wave owns (tile 0 owns its first lanes as well)
    using P = Pack<T, V>;
    ArgBlock2P<T> ap = (ArgBlock2P<T>)__builtin_amdgcn_kernarg_segment_ptr();
    const int lane = threadIdx.x & (kLanes - 1);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned bid = a.remap > 1 ? xcd_grouped(blockIdx.x, gridDim.x, (unsigned)a.remap) : blockIdx.x;
    const int ti = (int)(bid % a.nbx) * kPairTiles2D + w, bz = (int)(bid / a.nbx);
    if (ti >= a.n_tiles) return;                         // (no barrier in this kernel)
    const int tx = __builtin_amdgcn_readfirstlane(a.tiles[ti]);
    const int c0 = tx * OWN, i0 = c0 + lane * V;
    const int px = a.px, Nz = a.Nz, nz1 = a.nz1;
    const bool valid0 = i0 < px;                         // the last tile may reach past the row
    const bool own_lane0 = valid0 && lane < kLanes - HL && (lane >= HL || tx == 0);
    const unsigned lo = valid0 ? lane * V : 0;           // lanes past the row re-read the tile's first pack and are zeroed
    const bool first0 = i0 == 0;
    const bool in_n0 = valid0 && a.np[0][0] > 0 && i0 < a.xwn;
    const bool in_p0 = valid0 && a.np[0][1] > 0 && i0 >= a.xp0 && i0 < a.xp0 + a.xwp;
    P xCe = ldg<T, V>(a.xrow + 2 * (size_t)px + c0, lo), xCm = ldg<T, V>(a.xrow + 5 * (size_t)px + c0, lo);
    if (!valid0) { xCe = zero_pack<T, V>(); xCm = xCe; }
    const T* face_z = a.tab + (size_t)(1 * 2) * 6 * kMaxLayers;
    const int kb = bz * a.chunk, ke = min(kb + a.chunk, nz1);
    if (kb >= nz1) return;
    const bool src_lane0 = valid0 && a.iJ >= i0 && a.iJ < i0 + V;
    const T dt = a.dt, cz = a.dtd[1];
    // A wave none of whose cells, in any of the four stages, meets a wall, a CPML layer, the end of the row or the source
    // row runs the loop compiled with INNER: every "does this term exist here" test is true there and every zone the plain
    // one, so the tests and the selects the compiler makes of them are gone.
    const bool inner_wave = tx > 0 && c0 + kLanes * V <= px && (a.np[0][0] == 0 || c0 >= a.xwn) && (a.np[0][1] == 0 || c0 + kLanes * V <= a.xp0) &&
                            kb >= max(2, a.np[1][0] + 2) && ke <= Nz - a.np[1][1] - 2 && !(a.src_on && a.kJ >= kb - 1 && a.kJ <= ke + 1);
    auto run = [&](auto inner_tag) {
    constexpr bool INNER = decltype(inner_tag)::value;
    const bool valid = INNER || valid0, own_lane = INNER ? (lane >= HL && lane < kLanes - HL) : own_lane0;
    const bool first = !INNER && first0, in_n = !INNER && in_n0, in_p = !INNER && in_p0, src_lane = !INNER && src_lane0;

    auto flag_of = [&](int k) -> bool { return LZ && k >= 0 && k < nz1 && __builtin_amdgcn_readfirstlane((int)a.disp[(size_t)k * a.ntx + tx]) != 0; };
    auto layer = [](int zone) -> bool { return zone == Z_NEG || zone == Z_POS; };
    // one z term (the zone is the row's, the same for the whole wave): plain between the layers, or inside them
    //   phi = b*phi + a*curl;  d += C*curl + dt*phi   (FDTD2D.cpp:300-316 pattern) with the state handed in and out through `ph`
    auto zterm = [&](P& d, const P& curl, int zone, int l, int kb3, P& ph) {
        if (!INNER && zone == Z_NONE) return;
        const T d0 = d.v[0];
        if (INNER || zone == Z_MID) {
#pragma unroll
            for (int e = 0; e < V; ++e) d.v[e] = d.v[e] + cz * curl.v[e];
        } else {
            const FDTD_AS(4) T* t = (const FDTD_AS(4) T*)(face_z + (zone == Z_POS ? 6 * kMaxLayers : 0) + kb3 * kMaxLayers + l);
            const T b = t[0], aa = t[kMaxLayers], C = t[2 * kMaxLayers];
            const T p0 = ph.v[0];
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const T p = b * ph.v[e] + aa * curl.v[e];
                ph.v[e] = p;
                d.v[e] = d.v[e] + (C * curl.v[e] + dt * p);
            }
            if (first) ph.v[0] = p0;
        }
        if (first) d.v[0] = d0;
    };
    // one x term: every column carries its own coefficients (CPML columns the tabulated ones, plain columns (0, 0, dt/dx),
    // columns the reference's sweeps skip zeros); b and a are only fetched by lanes inside a strip
    auto xterm = [&](P& d, const P& curl, const P& xC, const T* xr_b, P& ph) {
        if (in_n || in_p) {
            const P b = ldg<T, V>(xr_b, lo), aa = ldg<T, V>(xr_b + px, lo);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const T p = b.v[e] * ph.v[e] + aa.v[e] * curl.v[e];
                ph.v[e] = p;
                d.v[e] = d.v[e] - (xC.v[e] * curl.v[e] + dt * p);
            }
        } else {
#pragma unroll
            for (int e = 0; e < V; ++e) d.v[e] = d.v[e] - xC.v[e] * curl.v[e];
        }
    };
    auto ld_x = [&](Slots<T> in, size_t off_n, ptrdiff_t off_p) -> P {
        P ph = zero_pack<T, V>();
        if (in_n) ph = ldg<T, V>(in[0] + off_n, lo);
        else if (in_p) ph = ldg<T, V>(in[1] + off_p, lo);
        return ph;
    };
    auto st_x = [&](Slots<T> out, size_t off_n, ptrdiff_t off_p, const P& ph) {
        if (in_n) stg<T, V>(out[0] + off_n, lo, ph);
        else if (in_p) stg<T, V>(out[1] + off_p, lo, ph);
    };

    // ---- state handed from iteration to iteration
    P hx0_km = zero_pack<T, V>();                                    // H0x(k-1)
    if (INNER || kb - 2 >= 0) {
        const size_t q = (size_t)(kb - 2) * px + c0;
        if constexpr (LZ) hx0_km = field_of<T, V>(ldg<T, V>(ap->Bx + q, lo), a.mu0, flag_of(kb - 2), ap->F1[1] + q, ap->C1 + q, ap->C2 + q, ap->C3 + q, lo);
        else hx0_km = divide<T, V>(ldg<T, V>(ap->Bx + q, lo), a.dm);
        if (!valid) hx0_km = zero_pack<T, V>();
    }
    P e1_km = zero_pack<T, V>(), d1_km = e1_km, bx0_km = e1_km, bz0_km = e1_km, s1e_km = e1_km, pez_km = e1_km, pex_km = e1_km;      // row k-1 after the first step's D half
    P h1x_km2 = e1_km, h1z_km2 = e1_km, e2_km2 = e1_km, b1x_km2 = e1_km, b1z_km2 = e1_km, sh1x_km2 = e1_km, sh1z_km2 = e1_km, pmz_km2 = e1_km, pmx_km2 = e1_km;   // row k-2
    bool fl_km = flag_of(kb - 2), fl_km2 = false;
    // "any Lorentz cell in this row segment": fetched one row ahead, with the rows themselves, so that no iteration starts
    // by waiting for its flag
    unsigned char pf = 0;
    auto request_flag = [&](int k) { if constexpr (LZ) pf = (k >= 0 && k < nz1) ? a.disp[(size_t)k * a.ntx + tx] : (unsigned char)0; };
    request_flag(kb - 1);
    // Dy, Bx, Bz of the row the next iteration starts with
    P nd = zero_pack<T, V>(), nbx = nd, nbz = nd, nacc = nd, acc1_km = nd;
    auto request_row = [&](int k) {
        if (INNER || (k >= 0 && k < nz1)) {
            const size_t q = (size_t)k * px + c0;
            nd = ldg<T, V>(ap->Dy + q, lo); nbx = ldg<T, V>(ap->Bx + q, lo); nbz = ldg<T, V>(ap->Bz + q, lo);
            if constexpr (ACC) { if (k >= kb && k < ke && !(a.acc_flags & 1)) nacc = ldg<T, V>(ap->acc + q, lo); }
        }
    };
    request_row(kb - 1);

    for (int k = kb - 1; k <= ke + 1; ++k) {
        if constexpr (LZ) asm volatile("" : "+s"(ap));
        const bool fl_k = LZ && __builtin_amdgcn_readfirstlane((int)pf) != 0;
        const size_t ro = (size_t)(INNER ? k : (k < 0 ? 0 : (k < nz1 ? k : nz1 - 1))) * px + c0;        // uniform offset of row k's segment (clamped: rows outside are not touched)
        const int km = k - 1, km2 = k - 2;
        const P cur_d = nd, cur_bx = nbx, cur_bz = nbz, cur_acc = nacc;
        if (k < ke + 1) { request_row(k + 1); request_flag(k + 1); }

        // ---------------------------------------------------------------- A: row k, first step: H0, Dy1, Ey1
        P d1 = zero_pack<T, V>(), e1 = d1, hx0 = d1, bx0 = d1, bz0 = d1, s1e = d1, pez = d1, pex = d1;
        if (INNER || (k >= 0 && k < nz1)) {
            d1 = cur_d; bx0 = cur_bx; bz0 = cur_bz;
            if (!valid) { d1 = zero_pack<T, V>(); bx0 = d1; bz0 = d1; }
            P hz0;
            if constexpr (LZ) {
                hx0 = field_of<T, V>(bx0, a.mu0, fl_k, ap->F1[1] + ro, ap->C1 + ro, ap->C2 + ro, ap->C3 + ro, lo);
                hz0 = field_of<T, V>(bz0, a.mu0, fl_k, ap->F1[2] + ro, ap->C1 + ro, ap->C2 + ro, ap->C3 + ro, lo);
            } else {
                divide2<T, V>(bx0, bz0, a.dm, hx0, hz0);
            }
            if (!valid) { hx0 = zero_pack<T, V>(); hz0 = hx0; }
            T hz_l = lane_below(hz0.v[V - 1]);
            if (!INNER && lane == 0) hz_l = (T)0;        // tile 0: the wall; other tiles: lane 0 is a halo pack whose first column nobody uses
            int lk;
            const int zk = INNER ? (lk = 0, (int)Z_MID) : zone_e(k, Nz, a.np[1][0], a.np[1][1], lk);
            if (!INNER && layer(zk)) pez = ldg<T, V>(ap->pe_z[zk == Z_POS] + ((size_t)lk * px + c0), lo);
            zterm(d1, diff<T, V>(hx0, hx0_km), zk, lk, 0, pez);
            if (INNER || (k >= 1 && k < Nz)) {
                if (!INNER) pex = ld_x(ap->pe_x, (size_t)k * a.xwn + c0, (ptrdiff_t)((size_t)k * a.xwp) + (c0 - a.xp0));
                xterm(d1, diff_back_x<T, V>(hz0, hz_l), xCe, a.xrow + c0, pex);
            }
            if (!INNER && (a.src_on & 1) && k == a.kJ && src_lane) {
#pragma unroll
                for (int e = 0; e < V; ++e) if (i0 + e == a.iJ) d1.v[e] = d1.v[e] - a.jy;
            }
            if (INNER || k < Nz) {
                if (LZ && fl_k) {
                    const P c1 = ldg<T, V>(ap->C1 + ro, lo), c2 = ldg<T, V>(ap->C2 + ro, lo), c3 = ldg<T, V>(ap->C3 + ro, lo);
                    const P f1 = ldg<T, V>(ap->F1[0] + ro, lo), f2 = ldg<T, V>(ap->F2[0] + ro, lo), s1 = ldg<T, V>(ap->S1[0] + ro, lo), s2 = ldg<T, V>(ap->S2[0] + ro, lo);
#pragma unroll
                    for (int e = 0; e < V; ++e) lorentz<T>(d1.v[e] / a.eps0, c1.v[e], c2.v[e], c3.v[e], f1.v[e], f2.v[e], s1.v[e], s2.v[e], e1.v[e], s1e.v[e]);
                    if (own_lane && k >= kb && k < ke) { stg<T, V>(ap->FX[0] + ro, lo, e1); stg<T, V>(ap->SX[0] + ro, lo, s1e); }
                } else {
                    e1 = divide<T, V>(d1, a.de);
                }
                if (!valid) e1 = zero_pack<T, V>();
            }
        }
        P acc1 = zero_pack<T, V>();
        if constexpr (ACC) {
            if (k >= kb && k < ke && k < Nz) {            // the window after the first step, own rows
                if (!(a.acc_flags & 1)) acc1 = cur_acc;
#pragma unroll
                for (int e = 0; e < V; ++e) acc1.v[e] = acc1.v[e] + e1.v[e] * e1.v[e] * a.acc_w;
                if ((a.acc_flags & 4) && own_lane) stg<T, V>(ap->acc_snap + ro, lo, acc1);
            }
        }

        // ---------------------------------------------------------------- B: row k-1, first step: B1, H1
        const T e1_r = lane_above(e1_km.v[0]);            // (taken with all lanes active)
        P b1x = bx0_km, b1z = bz0_km, h1x = zero_pack<T, V>(), h1z = h1x, sh1x = h1x, sh1z = h1x, pmz = h1x, pmx = h1x;
        if (k >= kb && (INNER || (km >= 0 && km < Nz))) {
            if constexpr (LZ) asm volatile("" : "+s"(ap));
            const size_t q = (size_t)km * px + c0;
            int lm;
            const int zm = INNER ? (lm = 0, (int)Z_MID) : zone_h(km, Nz, a.np[1][0], a.np[1][1], false, lm);
            if (!INNER && layer(zm)) pmz = ldg<T, V>(ap->pm_z[zm == Z_POS] + ((size_t)lm * px + c0), lo);
            zterm(b1x, diff<T, V>(e1, e1_km), zm, lm, 3, pmz);
            if (INNER || km >= 1) {
                if (!INNER) pmx = ld_x(ap->pm_x, (size_t)km * a.xwn + c0, (ptrdiff_t)((size_t)km * a.xwp) + (c0 - a.xp0));
                xterm(b1z, diff_fwd_x<T, V>(e1_km, e1_r), xCm, a.xrow + 3 * (size_t)px + c0, pmx);
            }
            if (LZ && fl_km) {
                const P c1 = ldg<T, V>(ap->C1 + q, lo), c2 = ldg<T, V>(ap->C2 + q, lo), c3 = ldg<T, V>(ap->C3 + q, lo);
                {
                    const P f1 = ldg<T, V>(ap->F1[1] + q, lo), f2 = ldg<T, V>(ap->F2[1] + q, lo), s1 = ldg<T, V>(ap->S1[1] + q, lo), s2 = ldg<T, V>(ap->S2[1] + q, lo);
#pragma unroll
                    for (int e = 0; e < V; ++e) lorentz<T>(b1x.v[e] / a.mu0, c1.v[e], c2.v[e], c3.v[e], f1.v[e], f2.v[e], s1.v[e], s2.v[e], h1x.v[e], sh1x.v[e]);
                }
                {
                    const P f1 = ldg<T, V>(ap->F1[2] + q, lo), f2 = ldg<T, V>(ap->F2[2] + q, lo), s1 = ldg<T, V>(ap->S1[2] + q, lo), s2 = ldg<T, V>(ap->S2[2] + q, lo);
#pragma unroll
                    for (int e = 0; e < V; ++e) lorentz<T>(b1z.v[e] / a.mu0, c1.v[e], c2.v[e], c3.v[e], f1.v[e], f2.v[e], s1.v[e], s2.v[e], h1z.v[e], sh1z.v[e]);
                }
                if (own_lane && km >= kb && km < ke) {
                    stg<T, V>(ap->FX[1] + q, lo, h1x); stg<T, V>(ap->SX[1] + q, lo, sh1x);
                    stg<T, V>(ap->FX[2] + q, lo, h1z); stg<T, V>(ap->SX[2] + q, lo, sh1z);
                }
            } else {
                divide2<T, V>(b1x, b1z, a.dm, h1x, h1z);
            }
            if (!valid) { h1x = zero_pack<T, V>(); h1z = h1x; }
        }

        // ---------------------------------------------------------------- C: row k-1, second step: Dy2, Ey2
        P e2 = zero_pack<T, V>();
        {
            T hz_l = lane_below(h1z.v[V - 1]);
            if (!INNER && lane == 0) hz_l = (T)0;
            if (k >= kb + 1 && (INNER || km < nz1)) {
                if constexpr (LZ) asm volatile("" : "+s"(ap));
                const size_t q = (size_t)km * px + c0;
                const bool st = own_lane && km >= kb && km < ke;
                P d2 = d1_km;
                int lk;
                const int zk = INNER ? (lk = 0, (int)Z_MID) : zone_e(km, Nz, a.np[1][0], a.np[1][1], lk);
                zterm(d2, diff<T, V>(h1x, h1x_km2), zk, lk, 0, pez_km);
                if (!INNER && layer(zk) && st) stg<T, V>(ap->pe_zn[zk == Z_POS] + ((size_t)lk * px + c0), lo, pez_km);
                if (INNER || (km >= 1 && km < Nz)) {
                    xterm(d2, diff_back_x<T, V>(h1z, hz_l), xCe, a.xrow + c0, pex_km);
                    if (!INNER && st) st_x(ap->pe_xn, (size_t)km * a.xwn + c0, (ptrdiff_t)((size_t)km * a.xwp) + (c0 - a.xp0), pex_km);
                }
                if (!INNER && (a.src_on & 2) && km == a.kJ && src_lane) {
#pragma unroll
                    for (int e = 0; e < V; ++e) if (i0 + e == a.iJ) d2.v[e] = d2.v[e] - a.jy2;
                }
                if (st) stg_nt<T, V>(ap->Dyn + q, lo, d2);
                if (INNER || km < Nz) {
                    if (LZ && fl_km) {
                        const P c1 = ldg<T, V>(ap->C1 + q, lo), c2 = ldg<T, V>(ap->C2 + q, lo), c3 = ldg<T, V>(ap->C3 + q, lo);
                        const P f2 = ldg<T, V>(ap->F1[0] + q, lo), s2 = ldg<T, V>(ap->S1[0] + q, lo);
                        P s2e;
#pragma unroll
                        for (int e = 0; e < V; ++e) lorentz<T>(d2.v[e] / a.eps0, c1.v[e], c2.v[e], c3.v[e], e1_km.v[e], f2.v[e], s1e_km.v[e], s2.v[e], e2.v[e], s2e.v[e]);
                        if (st) { stg<T, V>(ap->FY[0] + q, lo, e2); stg<T, V>(ap->SY[0] + q, lo, s2e); }
                    } else {
                        e2 = divide<T, V>(d2, a.de);
                    }
                    if (!valid) e2 = zero_pack<T, V>();
                    if constexpr (ACC) {
                        if (st) {                          // the window after the second step
                            P a2 = (a.acc_flags & 2) ? zero_pack<T, V>() : acc1_km;
#pragma unroll
                            for (int e = 0; e < V; ++e) a2.v[e] = a2.v[e] + e2.v[e] * e2.v[e] * a.acc_w;
                            stg<T, V>(ap->acc + q, lo, a2);
                        }
                    }
                }
            }
        }

        // ---------------------------------------------------------------- D: row k-2, second step: B2, H2
        const T e2_r = lane_above(e2_km2.v[0]);
        if (km2 >= kb && km2 < ke && (INNER || km2 < Nz)) {
            if constexpr (LZ) asm volatile("" : "+s"(ap));
            const size_t q = (size_t)km2 * px + c0;
            int lm;
            const int zm = INNER ? (lm = 0, (int)Z_MID) : zone_h(km2, Nz, a.np[1][0], a.np[1][1], false, lm);
            P b2x = b1x_km2, b2z = b1z_km2;
            zterm(b2x, diff<T, V>(e2, e2_km2), zm, lm, 3, pmz_km2);
            if (!INNER && layer(zm) && own_lane) stg<T, V>(ap->pm_zn[zm == Z_POS] + ((size_t)lm * px + c0), lo, pmz_km2);
            if (INNER || km2 >= 1) {
                xterm(b2z, diff_fwd_x<T, V>(e2_km2, e2_r), xCm, a.xrow + 3 * (size_t)px + c0, pmx_km2);
                if (!INNER && own_lane) st_x(ap->pm_xn, (size_t)km2 * a.xwn + c0, (ptrdiff_t)((size_t)km2 * a.xwp) + (c0 - a.xp0), pmx_km2);
            }
            if (own_lane) { stg_nt<T, V>(ap->Bxn + q, lo, b2x); stg_nt<T, V>(ap->Bzn + q, lo, b2z); }
            if (LZ && fl_km2) {
                const P c1 = ldg<T, V>(ap->C1 + q, lo), c2 = ldg<T, V>(ap->C2 + q, lo), c3 = ldg<T, V>(ap->C3 + q, lo);
                P fn, sn;
                {
                    const P f2 = ldg<T, V>(ap->F1[1] + q, lo), s2 = ldg<T, V>(ap->S1[1] + q, lo);
#pragma unroll
                    for (int e = 0; e < V; ++e) lorentz<T>(b2x.v[e] / a.mu0, c1.v[e], c2.v[e], c3.v[e], h1x_km2.v[e], f2.v[e], sh1x_km2.v[e], s2.v[e], fn.v[e], sn.v[e]);
                    if (own_lane) { stg<T, V>(ap->FY[1] + q, lo, fn); stg<T, V>(ap->SY[1] + q, lo, sn); }
                }
                {
                    const P f2 = ldg<T, V>(ap->F1[2] + q, lo), s2 = ldg<T, V>(ap->S1[2] + q, lo);
#pragma unroll
                    for (int e = 0; e < V; ++e) lorentz<T>(b2z.v[e] / a.mu0, c1.v[e], c2.v[e], c3.v[e], h1z_km2.v[e], f2.v[e], sh1z_km2.v[e], s2.v[e], fn.v[e], sn.v[e]);
                    if (own_lane) { stg<T, V>(ap->FY[2] + q, lo, fn); stg<T, V>(ap->SY[2] + q, lo, sn); }
                }
            }
        }

        // ---------------------------------------------------------------- hand-over
        hx0_km = hx0;
        h1x_km2 = h1x; h1z_km2 = h1z; e2_km2 = e2; b1x_km2 = b1x; b1z_km2 = b1z; sh1x_km2 = sh1x; sh1z_km2 = sh1z; pmz_km2 = pmz; pmx_km2 = pmx;
        e1_km = e1; d1_km = d1; bx0_km = bx0; bz0_km = bz0; s1e_km = s1e; pez_km = pez; pex_km = pex;
        fl_km2 = fl_km; fl_km = fl_k;
        acc1_km = acc1;
    }
    };
    if (inner_wave) run(std::true_type{}); else run(std::false_type{});
}

}  // namespace fdtd
