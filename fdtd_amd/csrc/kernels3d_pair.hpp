// kernels3d_pair.hpp -- two time steps of the 3-D update in one pass over memory.
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
// CPML: a cell's auxiliary state is advanced twice as well.  Between its two steps it waits in a per-wave
// piece of device memory (`pair_scr`): written in stage A/B, read back one iteration later in stage C/D, so it
// normally never leaves L2; both the D-side and the B-side state ping-pong between two buffers, because halo
// cells recompute their first step from the OLD state after the owner may have stored the new one.
//
// Restriction (checked by the host): the whole grid on one device.
#pragma once
#include <type_traits>
#include "kernels3d_fused.hpp"

namespace fdtd {

// (also used for plain global addresses: a V-element vector access aligned to V elements)
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

// One term along an axis whose zone is the same for the whole wave (y: the wave's row, z: the plane of this iteration):
// the plain update between the CPML layers, or inside them  phi = b*phi + a*curl;  d +- (C*curl + dt*phi)
// (FDTD3D.cpp:545-548 pattern) with the state handed in and out through `ph`.  `t` points at the (b, a, C) tables
// of this layer on the low side of the axis; the high side follows 6 tables further.
template <int SIGN, typename T, int V>
__device__ __forceinline__ void pair_cpml_u(Pack<T, V>& d, const Pack<T, V>& curl, bool skip0, bool high, const T* __restrict__ t, T dt, Pack<T, V>& ph) {
    // the tables are read through the scalar cache (the layer is the same for the whole wave): a vector load here would
    // make the term wait for every load the wave has in flight
    const FDTD_AS(4) T* tt = (const FDTD_AS(4) T*)(t + (high ? 6 * kMaxLayers : 0));
    const T b = tt[0], a = tt[kMaxLayers], C = tt[2 * kMaxLayers];
    const T d0 = d.v[0], p0 = ph.v[0];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        const T p = b * ph.v[e] + a * curl.v[e];
        ph.v[e] = p;
        const T v = C * curl.v[e] + dt * p;
        d.v[e] = SIGN > 0 ? d.v[e] + v : d.v[e] - v;
    }
    if (skip0) { ph.v[0] = p0; d.v[0] = d0; }
}
// The same along x, where every column has its own (b, a, C)
template <int SIGN, typename T, int V>
__device__ __forceinline__ void pair_cpml_x(Pack<T, V>& d, const Pack<T, V>& curl, const Pack<T, V>& b, const Pack<T, V>& a, const Pack<T, V>& C, T dt, Pack<T, V>& ph) {
#pragma unroll
    for (int e = 0; e < V; ++e) {
        const T p = b.v[e] * ph.v[e] + a.v[e] * curl.v[e];
        ph.v[e] = p;
        const T v = C.v[e] * curl.v[e] + dt * p;
        d.v[e] = SIGN > 0 ? d.v[e] + v : d.v[e] - v;
    }
}

// ACC: the launch also advances the running RMS window of the cell-centred |E|^2 (FDTD3D::run, FDTD3D.cpp:1074-1097)
// by both of its steps: acc(cell) (+)= (1/16)((sum4 Ex)^2 + (sum4 Ey)^2 + (sum4 Ez)^2) * weight after each step, in the
// reference's order of additions, from E1 and E2 while they are on chip.  The rows of the last THREE planes then stay in
// LDS (instead of two), so that a wave finds the row above of two consecutive planes there after the barrier.
template <typename T, bool MU, bool PML, int V, bool ACC = false>
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
    const int px = a.px, Ny = a.Ny, Nz = a.Nz;
    // x tiles: in a row (tile bx starts at column bx * XS), or -- wrap_l > 0 -- tile 0 holds both ends of the row: its
    // lanes [0, wrap_l) the first columns, the others the columns from wrap_r0 on; the tiles after it follow the left
    // part.  The seam inside tile 0 is a tile boundary like any other: two columns either side of it are computed, not
    // stored (the left part's last two are the next tile's, the right part's first two the last tile's).  The host makes
    // both parts reach two columns beyond the CPML strips, so that no other tile meets them.
    const bool wrap = a.wrap_l > 0 && bx == 0;
    const int c0 = wrap ? 0 : (a.wrap_l > 0 ? a.wrap_l * V - 4 + (bx - 1) * XS : bx * XS);      // base of this tile's addresses
    const int col = wrap ? ((int)lane < a.wrap_l ? (int)lo : a.wrap_r0 + ((int)lane - a.wrap_l) * V) : c0 + (int)lo;
    const bool live = col < px;
    // lanes past the end of the row read the row's last pack instead: whatever they compute is neither stored nor
    // seen by a live lane (the column right of the last stored one is the wall: zero, below)
    const int i0 = live ? col : px - V;
    const unsigned lc = (unsigned)(i0 - c0);             // this lane's columns relative to the tile's base
    const bool own_x = live && (wrap ? ((int)lane < a.wrap_l ? col + V <= a.wrap_l * V - 2 : col >= a.wrap_r0 + 2)
                                     : ((c0 == 0 || col >= c0 + 2) && col + V <= c0 + TW - 2 && (a.wrap_l == 0 || col < a.wrap_r0 + 2)));
    const bool wall_r0 = wrap ? lane == kLanes - 1 : col + V >= px;     // (the right part ends in the zero padding, or at the wall)
    const bool first0 = i0 == 0;
    const int jr = by * (W - 3) - 1 + w;
    const bool row_ok = jr >= 0 && jr < a.ny1;
    const int j = jr < 0 ? 0 : (jr < a.ny1 ? jr : a.ny1 - 1);
    const bool own = own_x && row_ok && w >= 1 && w < W - 2;
    // planes [kl0, zcut[0]) | [zcut[0], zcut[1]) | [zcut[1], kl1) are cut into chunks separately, so that the chunks that
    // touch the z layers are short ones of their own
    int kb, ke;
    if (bz < a.znb[0]) { kb = a.kl0 + bz * a.zchunk[0]; ke = min(kb + a.zchunk[0], a.zcut[0]); }
    else if (bz < a.znb[0] + a.znb[1]) { kb = a.zcut[0] + (bz - a.znb[0]) * a.zchunk[1]; ke = min(kb + a.zchunk[1], a.zcut[1]); }
    else { kb = a.zcut[1] + (bz - a.znb[0] - a.znb[1]) * a.zchunk[2]; ke = min(kb + a.zchunk[2], a.kl1); }
    const size_t sz = (size_t)a.ny1 * px;
    const bool row_lt0 = j < Ny, row_in0 = j >= 1 && j < Ny;
    const T cy = a.dtd[1], cz = a.dtd[2], dt = a.dt;
    // A workgroup none of whose cells, in any of the four stages, comes near a CPML layer runs the loop compiled
    // without the layer code (the straight-line loop of a grid without CPML); the others the general one.
    bool near_x = false, near_y = false, near_z = false;
    if constexpr (PML) {
        const int r0 = by * (W - 3) - 1;
        near_x = a.wrap_l > 0 ? wrap : ((a.np[0][0] > 0 && c0 < a.xwn) || (a.np[0][1] > 0 && c0 + TW > a.xp0));
        near_y = (a.np[1][0] > 0 && r0 <= a.np[1][0]) || (a.np[1][1] > 0 && r0 + W - 1 >= Ny - a.np[1][1]);
        const int g_lo = kb - 4 + a.zoff + 1, g_hi = ke + 1 + a.zoff;      // global planes the stages of this chunk touch
        near_z = (a.np[2][0] > 0 && g_lo <= a.np[2][0]) || (a.np[2][1] > 0 && g_hi >= Nz - a.np[2][1]);
    }
    // One loop per combination of layer families a workgroup can meet (x strips, y layers, z layers): each only carries
    // the code and the scalar state of its own families.
    // A wave whose cells all lie strictly inside the grid -- its row between the wall rows, its columns clear of both
    // walls and of the row's end, its planes (all four stages, every iteration) between the wall planes -- runs the
    // loop compiled with INNER: every "does this term exist here" test of the general loop is true there and every
    // zone the plain one, so the tests and the selects the compiler makes of them (52 v_cndmask, 23 branches and about
    // 80 scalar mask operations per iteration) are gone.  The waves of a workgroup may differ in this: both loops have
    // the same iterations and barriers.
    // (row Ny-1 and plane Nz-1 are not inner: the B update leaves the last cell along an axis out when its high face has no
    // CPML layers, FDTD3D.cpp:822-823)
    // (nor is a chunk whose last iterations, which only drain the pipeline, would address planes past the stored ones:
    // the top chunk of a slab below another one)
    const bool inner_wave = row_ok && j >= 1 && j < Ny - 1 && !wrap && c0 > 0 && c0 + TW < px && kb + a.zoff >= 2 && ke + 2 + a.zoff < Nz &&
                            ke + (ACC ? 2 : 1) < a.nzp;
    auto run = [&](auto lx_tag, auto ly_tag, auto lz_tag, auto inner_tag) {
        constexpr bool LX = PML && decltype(lx_tag)::value, LY = PML && decltype(ly_tag)::value, LZ = PML && decltype(lz_tag)::value;
        constexpr bool LAYERS = LX || LY || LZ;
        constexpr bool INNER = !LAYERS && decltype(inner_tag)::value;
        const bool row_lt = INNER || row_lt0, row_in = INNER || row_in0, first = !INNER && first0, wall_r = !INNER && wall_r0;
        const bool has_below = INNER || j >= 1;
        // a loop that carries exactly one family knows which one a non-zero family selector means
        constexpr int ONLY = (LX && !LY && !LZ) ? 1 : ((LY && !LX && !LZ) ? 2 : ((LZ && !LX && !LY) ? 3 : 0));
        auto fam_is = [](int fam, int q) -> bool { return ONLY ? (ONLY == q && fam != 0) : fam == q; };
        int lye, lyh;
        const int zye = INNER ? (lye = 0, (int)Z_MID) : zone_e(j, Ny, a.np[1][0], a.np[1][1], lye);
        const int zyh = INNER ? (lyh = 0, (int)Z_MID) : zone_h(j, Ny, a.np[1][0], a.np[1][1], true, lyh);
        const bool in_n = LX && a.np[0][0] > 0 && i0 < a.xwn;
        const bool in_p = LX && a.np[0][1] > 0 && i0 >= a.xp0 && i0 < a.xp0 + a.xwp;
        const bool wave_x = LX && __ballot(in_n || in_p) != 0;                       // some lane of this wave lies in an x strip
        // element offsets into the strip arrays: this row's share is fixed, a plane adds `xs_*` (32-bit: the host refuses
        // grids whose auxiliary arrays exceed 2^31 elements)
        const int xs_n = LX ? a.ny1 * a.xwn : 0, xs_p = LX ? a.ny1 * a.xwp : 0;
        const int xo_n = LX ? j * a.xwn + c0 : 0, xo_p = LX ? j * a.xwp + (c0 - a.xp0) : 0;
        const bool y_layer_e = LY && (zye == Z_NEG || zye == Z_POS), y_layer_h = LY && (zyh == Z_NEG || zyh == Z_POS);

        // LDS: [6][TW] x coefficients (b, a, C of the D side, then of the B side); then three regions
        // [2 planes][W rows][2 comps][TW]: E1 (x, z) for the row below, H1 (x, z) for the row above, E2 (x, z) for the row below
        // (ACC: three planes of E1 and of E2)
        constexpr int NB = ACC ? 3 : 2;
        T* const xs = smem;
        const size_t region = (size_t)2 * W * 2 * TW, region_e = (size_t)NB * W * 2 * TW;
        T* const e1s = smem + 6 * TW;
        T* const h1s = e1s + region_e;
        T* const e2s = h1s + region;
        for (int q = w; q < 6; q += W) lds_st<T, V>(xs + (size_t)q * TW + lo, ldg<T, V>(a.xrow + (size_t)q * px + c0, lc));
        __syncthreads();
        const P xcd = lds_ld<T, V>(xs + 2 * TW + lo), xch = lds_ld<T, V>(xs + 5 * TW + lo);
        auto slot = [&](T* reg, int par, int row) -> T* { return reg + ((size_t)(par * W + row) * 2) * TW + lo; };
        const int w_up = w + 1 < W ? w + 1 : w, w_dn = w >= 1 ? w - 1 : w;
        // What a halo row has to deliver decides which stages its wave runs: the row below the owned ones only H1 (stages
        // A, B), the first row above them E1, H1 and E2 (A, B, C), the second one only E1 (A).  Six of a workgroup's 64
        // stage executions per iteration are saved that way; the waves still meet at the barrier.
        const bool do_b = w <= W - 2, do_c = w >= 1 && w <= W - 2, do_d = w >= 1 && w <= W - 3;
        // CPML state between the two steps of a cell: 12 terms x 2 planes in flight, private to the wave, in device memory
        // (only touched inside the layers; the second step reads it back one iteration later, from L2)
        // (what only the layers need -- array extents, table and scratch pointers -- is read from the argument block
        // where it is used, not kept in scalar registers through the loop)
        const unsigned mid_off = (unsigned)(((size_t)blockIdx.x * W + w) * 24 * TW);
        auto mid_of = [&](int mpar, int term) -> T* { return ap->pair_scr + mid_off + (size_t)(mpar * 12 + term) * TW; };

        auto h_of = [&](const P& b, const P& m) -> P { return MU ? divide<T, V>(b, m) : divide<T, V>(b, a.mu0); };
        auto left_of_tile = [&](T shifted) -> T { return (!INNER && lane == 0 && c0 == 0) ? (T)0 : shifted; };
        auto right_of_tile = [&](T shifted) -> T { return wall_r ? (T)0 : shifted; };

        // ---- the two kinds of term.  step 0: state from the old buffers into `mid`; step 1: from `mid` into the new buffers.
        // Offsets into the CPML arrays are only formed inside the layers (most waves never get there).
        // zone-uniform axis (1 = y, 2 = z); kpc = local plane of the cell (y-face arrays are [plane][layer][column])
        auto uterm = [&](auto fast, auto sign, P& d, const P& curl, bool skip0, int zone, int l, auto axis_tag, int kb3, T cmid,
                         int step, Slots<T> src, Slots<T> dst, int kpc, int term, int mpar, bool st, bool have_pre, const P& pre) {
            constexpr int SIGN = decltype(sign)::value;
            constexpr bool FAST = decltype(fast)::value;          // the caller knows that no term of this stage lies in a layer
            constexpr int axis = decltype(axis_tag)::value;
            constexpr bool ON = axis == 2 ? LZ : LY;              // this workgroup can meet layers of this axis at all
            if (zone == Z_NONE) return;
            if (FAST || !ON || zone == Z_MID) { pair_axpy<SIGN, T, V>(d, cmid, curl, skip0); return; }
            if constexpr (ON && !FAST) {
                const int side = zone == Z_POS;
                const int off = axis == 2 ? (l * ap->ny1 + j) * px + c0 : (kpc * ap->np[1][side] + l) * px + c0;
                const T* face = ap->tab + (size_t)(axis * 2) * 6 * kMaxLayers;
                T* const m = mid_of(mpar, term);
                P ph = pre;
                if (!have_pre) ph = step == 0 ? ldg<T, V>(src[side] + off, lc) : ldg<T, V>(m, lo);
                pair_cpml_u<SIGN, T, V>(d, curl, skip0, side != 0, face + kb3 * kMaxLayers + l, dt, ph);
                if (step == 0) stg<T, V>(m, lo, ph);
                else if (st) stg<T, V>(dst[side] + off, lc, ph);
            }
        };
        // A wave waits for memory once per stage, not once per term: the state of ONE family of layers -- the x strips if
        // the wave has lanes in them, else the y layers if its row lies in one, else the z layers if the plane does -- is
        // requested ahead of the stage (with the field loads in step 0, right after the barrier in step 1); the terms of
        // another family that applies as well (edges and corners of the box) fetch theirs where they need it.
        auto family = [&](bool ylayer, int zk) -> int {
            return __builtin_amdgcn_readfirstlane((LX && wave_x) ? 1 : ((LY && ylayer) ? 2 : ((LZ && (zk == Z_NEG || zk == Z_POS)) ? 3 : 0)));
        };
        // the idx-th (0, 1) term of family `fam` on the D side (mag = false) or the B side: state of step 0 from the old
        // buffers (plane kpc; z layer lk of zone zk), of step 1 from `mid`
        auto request = [&](int fam, bool mag, int idx, int step, int kpc, int zk, int lk, int mpar) -> P {
            P ph = zero_pack<T, V>();
            if constexpr (LAYERS) {
                if (fam == 0) return ph;
                if (ONLY) fam = ONLY;
                const int axis = fam == 1 ? 0 : (fam == 2 ? 1 : 2);
                int comp, term;
                if (fam == 1) { comp = idx ? 2 : 1; term = mag ? (idx ? 11 : 8) : (idx ? 4 : 3); }
                else if (fam == 2) { comp = idx ? 2 : 0; term = mag ? (idx ? 10 : 7) : (idx ? 5 : 1); }
                else { comp = idx ? 1 : 0; term = mag ? (idx ? 9 : 6) : (idx ? 2 : 0); }
                if (step == 1) {
                    // (a plain per-lane address: with the term number only known at run time the compiler does not accept
                    // this base as wave-uniform)
                    const T* const m = mid_of(mpar, term) + lo;
                    if (fam != 1) ph = lds_ld<T, V>(m);
                    else if (in_n || in_p) ph = lds_ld<T, V>(m);
                    return ph;
                }
                Slots<T> src = mag ? (Slots<T>)ap->phM[comp][axis] : (Slots<T>)ap->phE[comp][axis];
                if (LX && fam == 1) {
                    const T* const pn = src[0] + (kpc * xs_n + xo_n);
                    const T* const pp = src[1] + (kpc * xs_p + xo_p);
                    if (in_n) ph = ldg<T, V>(pn, lc);
                    if (in_p) ph = ldg<T, V>(pp, lc);
                } else if (LY && fam == 2) {
                    const int side = (mag ? zyh : zye) == Z_POS, l = mag ? lyh : lye;
                    ph = ldg<T, V>(src[side] + ((kpc * ap->np[1][side] + l) * px + c0), lc);
                } else if (LZ && fam == 3) {
                    const int side = zk == Z_POS;
                    ph = ldg<T, V>(src[side] + ((lk * ap->ny1 + j) * px + c0), lc);
                }
            }
            return ph;
        };
        auto xterm = [&](auto fast, auto sign, P& d, const P& curl, int kb3, const P& xc, int step, P ph, Slots<T> dst, int kpc, int term, int mpar, bool st) {
            constexpr int SIGN = decltype(sign)::value;
            constexpr bool FAST = decltype(fast)::value;
            if (LX && !FAST && (in_n || in_p)) {
                if constexpr (LX && !FAST) {
                    const P b = lds_ld<T, V>(xs + (size_t)kb3 * TW + lo), aa = lds_ld<T, V>(xs + (size_t)(kb3 + 1) * TW + lo);
                    pair_cpml_x<SIGN, T, V>(d, curl, b, aa, xc, dt, ph);
                    if (step == 0) stg<T, V>(mid_of(mpar, term), lo, ph);
                    else if (st) {
                        if (in_n) stg<T, V>(dst[0] + (kpc * xs_n + xo_n), lc, ph); else stg<T, V>(dst[1] + (kpc * xs_p + xo_p), lc, ph);
                    }
                }
            } else {
                pair_axpy_x<SIGN, T, V>(d, xc, curl);
            }
        };
        using AxY = std::integral_constant<int, 1>;
        using AxZ = std::integral_constant<int, 2>;
        using Plus = std::integral_constant<int, +1>;
        using Minus = std::integral_constant<int, -1>;

        // D of plane (kp local, k global) from H of the same step: terms, order and index ranges of k3_fused
        auto update_d = [&](P* dd, const P& hx, const P& hy, const P& hz, const P& hx_m, const P& hy_m, const P& hx_jm, const P& hz_jm,
                            T hy_l, T hz_l, int kp, int k, int step, int mpar, bool st, int fam, const P& pre0, const P& pre1) {
            int lke;
            const int zke = INNER ? (lke = 0, (int)Z_MID) : zone_e(k, Nz, a.np[2][0], a.np[2][1], lke);
            const bool pl_lt = INNER || (k >= 0 && k < Nz), pl_in = INNER || (k >= 1 && k < Nz);
            const int kpc = kp < a.nzp ? kp : a.nzp - 1;
            auto terms = [&](auto fast) {
                if (row_lt) uterm(fast, Minus{}, dd[0], diff<T, V>(hy, hy_m), false, zke, lke, AxZ{}, 0, cz, step, ap->phE[0][2], ap->phEn[0][2], kpc, 0, mpar, st, fam_is(fam, 3), pre0);
                if (pl_in) uterm(fast, Plus{}, dd[0], diff<T, V>(hz, hz_jm), false, zye, lye, AxY{}, 0, cy, step, ap->phE[0][1], ap->phEn[0][1], kpc, 1, mpar, st, fam_is(fam, 2), pre0);
                if (((a.src_on >> step) & 1) && k == a.kJ) {
                    const P m = ldg<T, V>(a.mag + (size_t)j * px + c0, lc);
                    const T jx = step == 0 ? a.jx : a.jx2;
#pragma unroll
                    for (int e = 0; e < V; ++e) dd[0].v[e] = dd[0].v[e] + jx * m.v[e];
                }
                if (row_lt) {
                    uterm(fast, Plus{}, dd[1], diff<T, V>(hx, hx_m), first, zke, lke, AxZ{}, 0, cz, step, ap->phE[1][2], ap->phEn[1][2], kpc, 2, mpar, st, fam_is(fam, 3), pre1);
                    if (pl_in) xterm(fast, Minus{}, dd[1], diff_back_x<T, V>(hz, hz_l), 0, xcd, step, pre0, ap->phEn[1][0], kpc, 3, mpar, st);
                }
                if (pl_lt) {
                    if (row_in) xterm(fast, Plus{}, dd[2], diff_back_x<T, V>(hy, hy_l), 0, xcd, step, pre1, ap->phEn[2][0], kpc, 4, mpar, st);
                    uterm(fast, Minus{}, dd[2], diff<T, V>(hx, hx_jm), first, zye, lye, AxY{}, 0, cy, step, ap->phE[2][1], ap->phEn[2][1], kpc, 5, mpar, st, fam_is(fam, 2), pre1);
                }
            };
            // most waves are nowhere near a layer: they take the straight-line code of the kernel without CPML
            if constexpr (!LAYERS) terms(std::true_type{});
            else if (!wave_x && !y_layer_e && (!LZ || (zke != Z_NEG && zke != Z_POS))) terms(std::true_type{});
            else terms(std::false_type{});
        };
        // B of plane (kp local, km global) from E of the same step at planes km (`e`) and km+1 (`e_kp`), the row above
        // (`ex_jp`, `ez_jp`) and the column to the right (`ey_r`, `ez_r`)
        auto update_b = [&](P* bb, const P* e, const P* e_kp, const P& ex_jp, const P& ez_jp, T ey_r, T ez_r,
                            int kp, int km, int step, int mpar, bool st, int fam, const P& pre0, const P& pre1) {
            int lkh;
            const int zkh = INNER ? (lkh = 0, (int)Z_MID) : zone_h(km, Nz, a.np[2][0], a.np[2][1], true, lkh);
            const bool bl_lt = INNER || (km >= 0 && km < Nz), bl_in = INNER || (km >= 1 && km < Nz);
            const int kpc = kp < 0 ? 0 : (kp < a.nzp ? kp : a.nzp - 1);
            auto terms = [&](auto fast) {
                if (row_lt && bl_lt) {
                    uterm(fast, Plus{}, bb[0], diff<T, V>(e_kp[1], e[1]), first, zkh, lkh, AxZ{}, 3, cz, step, ap->phM[0][2], ap->phMn[0][2], kpc, 6, mpar, st, fam_is(fam, 3), pre0);
                    uterm(fast, Minus{}, bb[0], diff<T, V>(ez_jp, e[2]), first, zyh, lyh, AxY{}, 3, cy, step, ap->phM[0][1], ap->phMn[0][1], kpc, 7, mpar, st, fam_is(fam, 2), pre0);
                }
                if (bl_lt && row_in) {
                    xterm(fast, Plus{}, bb[1], diff_fwd_x<T, V>(e[2], ez_r), 3, xch, step, pre0, ap->phMn[1][0], kpc, 8, mpar, st);
                    uterm(fast, Minus{}, bb[1], diff<T, V>(e_kp[0], e[0]), false, zkh, lkh, AxZ{}, 3, cz, step, ap->phM[1][2], ap->phMn[1][2], kpc, 9, mpar, st, fam_is(fam, 3), pre1);
                }
                if (row_lt && bl_in) {
                    uterm(fast, Plus{}, bb[2], diff<T, V>(ex_jp, e[0]), false, zyh, lyh, AxY{}, 3, cy, step, ap->phM[2][1], ap->phMn[2][1], kpc, 10, mpar, st, fam_is(fam, 2), pre1);
                    xterm(fast, Minus{}, bb[2], diff_fwd_x<T, V>(e[1], ey_r), 3, xch, step, pre1, ap->phMn[2][0], kpc, 11, mpar, st);
                }
            };
            if constexpr (!LAYERS) terms(std::true_type{});
            else if (!wave_x && !y_layer_h && (!LZ || (zkh != Z_NEG && zkh != Z_POS))) terms(std::true_type{});
            else terms(std::false_type{});
        };

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
        // ACC: the window value of the cells one and two planes back after their first step, and Ey2(i) + Ey2(i+1) of the
        // plane whose own row has left the registers (the rest of that plane is read back from LDS)
        P acc_km1 = zero_pack<T, V>(), acc_km2 = zero_pack<T, V>(), ey2_pair = zero_pack<T, V>();
        int q3 = ACC ? (kb - 1) % 3 : 0;                             // ACC: which of the three LDS planes this iteration writes
        // a pack shifted one column to the left: element e becomes the value of column e + 1
        auto shl = [&](const P& x) -> P {
            P r;
#pragma unroll
            for (int e = 0; e + 1 < V; ++e) r.v[e] = x.v[e + 1];
            r.v[V - 1] = right_of_tile(lane_above(x.v[0]));
            return r;
        };
        // (1/16)((sum of four)^2 + ...), each sum left to right as FDTD3D.cpp:1075-1081
        auto center_sq = [&](const P& x0, const P& x1, const P& x2, const P& x3, const P& y01, const P& y2, const P& y3,
                             const P& z0, const P& z1, const P& z2, const P& z3) -> P {
            P r;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const T sx = ((x0.v[e] + x1.v[e]) + x2.v[e]) + x3.v[e];
                const T sy = (y01.v[e] + y2.v[e]) + y3.v[e];
                const T s_z = ((z0.v[e] + z1.v[e]) + z2.v[e]) + z3.v[e];
                r.v[e] = (T)0.0625 * ((sx * sx + sy * sy) + s_z * s_z);
            }
            return r;
        };

        for (int kl = kb - 1; kl <= ke + (ACC ? 2 : 1); ++kl) {
            asm volatile("" : "+s"(ap));
            // Issue priority by stage: a wave that is about to request its planes goes first on its SIMD, so that its loads
            // travel while the other waves do arithmetic; the stage in front of the next iteration's loads comes second
            // (measured at 1024^3 with CPML, four runs each: 11.10-11.14 ms per step against 11.20-11.27 without).
            __builtin_amdgcn_s_setprio(3);
            const int klc = (INNER || kl < a.nzp) ? kl : a.nzp - 1;          // iterations past the last plane only drain the pipeline
            const size_t ro = ((size_t)klc * a.ny1 + j) * px + c0;
            const int k = kl + a.zoff;
            const int par = kl & 1;
            // LDS planes of the E rows: written by this iteration, by the one before, by the one before that
            const int ecur = ACC ? q3 : par, eprev = ACC ? (q3 == 0 ? 2 : q3 - 1) : (par ^ 1), eprev2 = ACC ? (q3 == 2 ? 0 : q3 + 1) : 0;

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
            const size_t rj = has_below ? ro - px : ro;
            const P b0j = ldg<T, V>(ap->B[0] + rj, lc), b2j = ldg<T, V>(ap->B[2] + rj, lc);
            P m0j = zero_pack<T, V>(), m2j = zero_pack<T, V>();
            if (MU) { m0j = ldg<T, V>(ap->mu[0] + rj, lc); m2j = ldg<T, V>(ap->mu[2] + rj, lc); }
#pragma unroll
            for (int c = 0; c < 3; ++c) { d1[c] = ldg<T, V>(ap->D[c] + ro, lc); ep[c] = ldg<T, V>(ap->eps[c] + ro, lc); }
            // the CPML state this iteration's first-step terms start from (one family of layers per stage, see `family`)
            int lkA, lkB;
            const int zkA = zone_e(k, Nz, a.np[2][0], a.np[2][1], lkA), zkB = zone_h(k - 1, Nz, a.np[2][0], a.np[2][1], true, lkB);
            const int famA = LAYERS ? family(y_layer_e, zkA) : 0, famB = (LAYERS && do_b && kl >= kb) ? family(y_layer_h, zkB) : 0;
            const int klm = kl >= 1 ? (kl - 1 < a.nzp ? kl - 1 : a.nzp - 1) : 0;
            const P preA0 = request(famA, false, 0, 0, klc, zkA, lkA, par), preA1 = request(famA, false, 1, 0, klc, zkA, lkA, par);
            const P preB0 = request(famB, true, 0, 0, klm, zkB, lkB, par ^ 1), preB1 = request(famB, true, 1, 0, klm, zkB, lkB, par ^ 1);

            const P hx = h_of(bo[0], mu_[0]), hy = h_of(bo[1], mu_[1]), hz = h_of(bo[2], mu_[2]);
            P hx_jm = zero_pack<T, V>(), hz_jm = zero_pack<T, V>();
            if (has_below) { hx_jm = h_of(b0j, m0j); hz_jm = h_of(b2j, m2j); }
            {
                const T hy_l = left_of_tile(lane_below(hy.v[V - 1])), hz_l = left_of_tile(lane_below(hz.v[V - 1]));
                update_d(d1, hx, hy, hz, hx_km, hy_km, hx_jm, hz_jm, hy_l, hz_l, kl, k, 0, par, false, famA, preA0, preA1);
            }
            P e1[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) e1[c] = divide<T, V>(d1[c], ep[c]);
            lds_st<T, V>(slot(e1s, ecur, w), e1[0]); lds_st<T, V>(slot(e1s, ecur, w) + TW, e1[2]);

            // ------------------------------------------------------------ B: B1(k-1), H1(k-1)
            __builtin_amdgcn_s_setprio(0);
            P b1[3], h1[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) { b1[c] = b0_km[c]; h1[c] = zero_pack<T, V>(); }
            if (do_b && kl >= kb) {
                const T* up = slot(e1s, eprev, w_up);                // E1(k-1) of the row above, published one iteration ago
                const P ex_jp = lds_ld<T, V>(up), ez_jp = lds_ld<T, V>(up + TW);
                const T ey_r = right_of_tile(lane_above(e1_km[1].v[0])), ez_r = right_of_tile(lane_above(e1_km[2].v[0]));
                update_b(b1, e1_km, e1, ex_jp, ez_jp, ey_r, ez_r, kl - 1, k - 1, 0, par ^ 1, false, famB, preB0, preB1);
#pragma unroll
                for (int c = 0; c < 3; ++c) h1[c] = h_of(b1[c], mu_km[c]);
                lds_st<T, V>(slot(h1s, par, w), h1[0]); lds_st<T, V>(slot(h1s, par, w) + TW, h1[2]);
            }
            // the state the second-step terms of this iteration continue from (left in `mid` one iteration ago): requested
            // before the barrier, so that the wait for the other waves covers its way from L2
            int lkC, lkD;
            const int zkC = zone_e(k - 1, Nz, a.np[2][0], a.np[2][1], lkC), zkD = zone_h(k - 2, Nz, a.np[2][0], a.np[2][1], true, lkD);
            const int famC = (LAYERS && do_c && kl >= kb + 1) ? family(y_layer_e, zkC) : 0, famD = (LAYERS && do_d && kl >= kb + 2) ? family(y_layer_h, zkD) : 0;
            const P preC0 = request(famC, false, 0, 1, 0, zkC, lkC, par ^ 1), preC1 = request(famC, false, 1, 1, 0, zkC, lkC, par ^ 1);
            const P preD0 = request(famD, true, 0, 1, 0, zkD, lkD, par), preD1 = request(famD, true, 1, 1, 0, zkD, lkD, par);
            __syncthreads();

            // ------------------------------------------------------------ C: D2(k-1), E2(k-1)
            P e2[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) e2[c] = zero_pack<T, V>();
            if (do_c && kl >= kb + 1) {
                asm volatile("" : "+s"(ap));
                const T* dn = slot(h1s, par, w_dn);                  // H1(k-1) of the row below, published before the barrier
                P hx_jm1 = lds_ld<T, V>(dn), hz_jm1 = lds_ld<T, V>(dn + TW);
                if (!has_below) { hx_jm1 = zero_pack<T, V>(); hz_jm1 = zero_pack<T, V>(); }
                const T hy_l = left_of_tile(lane_below(h1[1].v[V - 1])), hz_l = left_of_tile(lane_below(h1[2].v[V - 1]));
                const int km = k - 1;
                const bool st = own && kl - 1 < ke;
                P d2[3] = {d1_km[0], d1_km[1], d1_km[2]};
                update_d(d2, h1[0], h1[1], h1[2], h1_km2[0], h1_km2[1], hx_jm1, hz_jm1, hy_l, hz_l, kl - 1, km, 1, par ^ 1, st, famC, preC0, preC1);
                if (st) {
                    const size_t rs = ro - sz * (size_t)(klc - (kl - 1));
                    stg_nt<T, V>(ap->Dn[0] + rs, lc, d2[0]);
                    if (row_lt) stg_nt<T, V>(ap->Dn[1] + rs, lc, d2[1]);
                    if (INNER || (km >= 0 && km < Nz)) stg_nt<T, V>(ap->Dn[2] + rs, lc, d2[2]);
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) e2[c] = divide<T, V>(d2[c], ep_km[c]);
                lds_st<T, V>(slot(e2s, ecur, w), e2[0]); lds_st<T, V>(slot(e2s, ecur, w) + TW, e2[2]);
            }

            // ------------------------------------------------------------ D: B2(k-2)
            __builtin_amdgcn_s_setprio(1);
            if (do_d && kl >= kb + 2) {
                const T* up = slot(e2s, eprev, w_up);                // E2(k-2) of the row above, published one iteration ago
                const P ex_jp = lds_ld<T, V>(up), ez_jp = lds_ld<T, V>(up + TW);
                const T ey_r = right_of_tile(lane_above(e2_km2[1].v[0])), ez_r = right_of_tile(lane_above(e2_km2[2].v[0]));
                const int km2 = k - 2;
                const bool bl_lt = INNER || (km2 >= 0 && km2 < Nz);
                P b2[3] = {b1_km2[0], b1_km2[1], b1_km2[2]};
                const bool st = own && (!ACC || kl - 2 < ke);        // (ACC: the loop runs one iteration longer)
                update_b(b2, e2_km2, e2, ex_jp, ez_jp, ey_r, ez_r, kl - 2, km2, 1, par, st, famD, preD0, preD1);
                if (st) {
                    const size_t rs = ro - sz * (size_t)(klc - (kl - 2));
                    if (row_lt && bl_lt) stg_nt<T, V>(ap->Bn[0] + rs, lc, b2[0]);
                    if (bl_lt) stg_nt<T, V>(ap->Bn[1] + rs, lc, b2[1]);
                    if (row_lt) stg_nt<T, V>(ap->Bn[2] + rs, lc, b2[2]);
                }
            }

            // ------------------------------------------------------------ ACC: the RMS window of both steps
            if constexpr (ACC) {
                asm volatile("" : "+s"(ap));
                const T wgt = ap->acc_w;
                const int flags = ap->acc_flags;          // 1: the first step opens a new window, 2: the second does, 4: keep the window as the first step leaves it
                // first step, cells between the planes k-1 and k: own rows in registers, the row above in LDS (plane k
                // published before this iteration's barrier, plane k-1 one iteration earlier)
                P a1 = zero_pack<T, V>();
                if (do_d && kl >= kb + 1) {
                    const T* u1 = slot(e1s, ecur, w_up);
                    const T* u0 = slot(e1s, eprev, w_up);
                    const P xu1 = lds_ld<T, V>(u1), xu0 = lds_ld<T, V>(u0), zu0 = lds_ld<T, V>(u0 + TW);
                    P y01;
                    {
                        const P r = shl(e1_km[1]);
#pragma unroll
                        for (int e = 0; e < V; ++e) y01.v[e] = e1_km[1].v[e] + r.v[e];
                    }
                    const P c = center_sq(e1_km[0], xu0, e1[0], xu1, y01, e1[1], shl(e1[1]), e1_km[2], shl(e1_km[2]), zu0, shl(zu0));
                    const int km = k - 1;
                    const bool on = own && kl - 1 < ke && row_lt && km >= 0 && km < Nz;
                    const size_t rs = ro - sz * (size_t)(klc - (kl - 1));
                    if (on && !(flags & 1)) a1 = ldg<T, V>(ap->acc + rs, lc);
#pragma unroll
                    for (int e = 0; e < V; ++e) a1.v[e] = a1.v[e] + c.v[e] * wgt;
                    if (on && (flags & 4)) stg<T, V>(ap->acc_snap + rs, lc, a1);
                }
                // second step, cells between the planes k-3 and k-2: E2(k-2) of the own row in registers, of the row above
                // in LDS; E2(k-3) in LDS (x, z) and in `ey2_pair` (y)
                if (do_d && kl >= kb + 3) {
                    const T* o3 = slot(e2s, eprev2, w);
                    const T* u3 = slot(e2s, eprev2, w_up);
                    const T* u2 = slot(e2s, eprev, w_up);
                    const P xo3 = lds_ld<T, V>(o3), zo3 = lds_ld<T, V>(o3 + TW), xu3 = lds_ld<T, V>(u3), zu3 = lds_ld<T, V>(u3 + TW), xu2 = lds_ld<T, V>(u2);
                    const P c = center_sq(xo3, xu3, e2_km2[0], xu2, ey2_pair, e2_km2[1], shl(e2_km2[1]), zo3, shl(zo3), zu3, shl(zu3));
                    const int km3 = k - 3;
                    const bool on = own && kl - 3 < ke && row_lt && km3 >= 0 && km3 < Nz;
                    const size_t rs = ((size_t)(kl - 3) * a.ny1 + j) * px + c0;
                    P out = acc_km2;
                    if (kl - 3 != ap->acc_defer_kl) {      // (a deferred layer keeps the window as its first step left it)
                        if (flags & 2) out = zero_pack<T, V>();
#pragma unroll
                        for (int e = 0; e < V; ++e) out.v[e] = out.v[e] + c.v[e] * wgt;
                    }
                    if (on) stg<T, V>(ap->acc + rs, lc, out);
                }
                {
                    const P r = shl(e2_km2[1]);
#pragma unroll
                    for (int e = 0; e < V; ++e) ey2_pair.v[e] = e2_km2[1].v[e] + r.v[e];
                }
                acc_km2 = acc_km1; acc_km1 = a1;
                q3 = q3 == 2 ? 0 : q3 + 1;
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
    };
    using No = std::false_type;
    using Yes = std::true_type;
    auto plain = [&]() { if (inner_wave) run(No{}, No{}, No{}, Yes{}); else run(No{}, No{}, No{}, No{}); };
    if constexpr (!PML) plain();
    else if (near_x) {
        if (near_y) { if (near_z) run(Yes{}, Yes{}, Yes{}, No{}); else run(Yes{}, Yes{}, No{}, No{}); }
        else { if (near_z) run(Yes{}, No{}, Yes{}, No{}); else run(Yes{}, No{}, No{}, No{}); }
    } else {
        if (near_y) { if (near_z) run(No{}, Yes{}, Yes{}, No{}); else run(No{}, Yes{}, No{}, No{}); }
        else { if (near_z) run(No{}, No{}, Yes{}, No{}); else plain(); }
    }
}

}  // namespace fdtd
