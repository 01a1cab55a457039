// kernels2d.hpp -- gfx950 kernels of the 2-D (Ey, Hx, Hz) update with the Lorentz recursion.
//
// Replaces FDTD2D::stepOnce (fdtd2d/src/FDTD2D.cpp:275-482).  Two kernels per step:
//   E part: Dy curl + CPML + line source, then the Lorentz update of Ey/SEy   (:289-379)
//   H part: Bx, Bz curl + CPML, then the Lorentz updates of Hx/SHx, Hz/SHz    (:383-481)
// The reference rotates its history arrays with a swap and a full copy per field; here the two
// history buffers of a field ping-pong: the new sample overwrites the n-2 sample of the same
// cell (the recursion is point-wise) and the host swaps the two pointers after the launch.
//
// Tiles (64 lanes of one row) in which C1 = C2 = C3 = 0 take a short cut: there the recursion is
// exactly f = drive / 1 with S = 0 (every product with a zero coefficient is a zero), so only the
// new sample is written and no history is read.  The per-tile flags are rebuilt on the device
// whenever a coefficient array is uploaded.
//
// Device layout: [row k][i], i unit-stride, rows padded to a multiple of 256 elements, Nz+1
// rows; the wall column i = Nx of Ey and Hx is never written by the reference and not stored.
// A wave owns 64 lanes x 16 bytes of one row and marches through a chunk of rows, carrying
// the z-neighbour in registers; a workgroup is four such waves on adjacent x tiles.
#pragma once
#include "kernels3d.hpp"

namespace fdtd {

template <typename T>
struct K2 {
    T* Dy; T* Bx; T* Bz;
    const T* F1[3]; T* F2[3];      // history n-1 (read) and n-2 (read, then overwritten with the new sample)
    const T* S1[3]; T* S2[3];
    const T* C1; const T* C2; const T* C3;
    T* ph_ey_z[2]; T* ph_ey_x[2]; T* ph_mx_z[2]; T* ph_mz_x[2];
    const T* tab;                  // [axis 2][side][kind 6][kMaxLayers], axis 0 = x, 1 = z
    const T* xrow;                 // [6][px]
    int Nx, Nz, px, nz1;
    int np[2][2];
    int xwn, xwp, xp0;
    int iJ, kJ, src_on;
    T jy;                          // dt * Jy[n]
    T dt, dtd[2], mu0, eps0;
    int chunk, nbx, nbz, ntx;      // rows per wave, workgroups along x and z, x tiles in total
    int remap;
    const unsigned char* disp;     // [nz1][ntx]: 1 where the 64-lane tile of that row holds a dispersive cell
};

// Second-order Lorentz recursion of one sample (FDTD2D.cpp:369-373):
//   num = drive - 2 C1 f1 - C1 f2 + C2 s1 + C3 s2;  f = num / (1 + C1)
//   s   = C1 f + 2 C1 f1 + C1 f2 - C2 s1 - C3 s2
template <typename T>
__device__ __forceinline__ void lorentz(T drive, T c1, T c2, T c3, T f1, T f2, T s1, T s2, T& f, T& s) {
    const T num = drive - (T)2.0 * c1 * f1 - c1 * f2 + c2 * s1 + c3 * s2;
    f = num / ((T)1.0 + c1);
    s = c1 * f + (T)2.0 * c1 * f1 + c1 * f2 - c2 * s1 - c3 * s2;
}

template <typename T>
struct Tile2 {
    int lane, i0, kb, ke, tx;
    bool ok;
    unsigned m_lt, m_in;
    bool lane_n, lane_p;
};
template <typename T>
__device__ __forceinline__ Tile2<T> decode_tile2(const K2<T>& a) {
    constexpr int V = VecWidth<T>::V;
    Tile2<T> t;
    t.lane = threadIdx.x & (kLanes - 1);
    const int w = threadIdx.x >> 6;
    const unsigned bid = a.remap == 1 ? xcd_contiguous(blockIdx.x, gridDim.x) : (a.remap > 1 ? xcd_grouped(blockIdx.x, gridDim.x, (unsigned)a.remap) : blockIdx.x);
    const int bx = bid % a.nbx, bz = bid / a.nbx;
    const int tx = bx * kWavesPerBlock + w;
    t.ok = tx < a.ntx;
    t.tx = tx;
    t.i0 = (tx * kLanes + t.lane) * V;
    t.kb = bz * a.chunk;
    t.ke = min(t.kb + a.chunk, a.nz1);
    t.m_lt = 0; t.m_in = 0;
#pragma unroll
    for (int e = 0; e < V; ++e) {
        const int i = t.i0 + e;
        if (i < a.Nx) t.m_lt |= 1u << e;
        if (i >= 1 && i < a.Nx) t.m_in |= 1u << e;
    }
    t.lane_n = a.np[0][0] > 0 && t.i0 < a.xwn;
    t.lane_p = a.np[0][1] > 0 && t.i0 >= a.xp0 && t.i0 < a.xp0 + a.xwp;
    return t;
}

template <typename T>
__global__ void __launch_bounds__(256) k2_part_e(const K2<T> a) {
    constexpr int V = VecWidth<T>::V;
    using P = Pack<T, V>;
    const Tile2<T> t = decode_tile2(a);
    if (!t.ok) return;
    const int i0 = t.i0, lane = t.lane, px = a.px, Nx = a.Nx, Nz = a.Nz;
    const T* __restrict__ Hx = a.F1[1]; const T* __restrict__ Hz = a.F1[2];
    const T* face_z = a.tab + (size_t)(1 * 2) * 6 * kMaxLayers;

    size_t idx = (size_t)t.kb * px + i0;
    P hx_km = zero_pack<T, V>();
    if (t.kb >= 1) hx_km = ldv<T, V>(Hx + idx - px);
    for (int k = t.kb; k < t.ke; ++k, idx += px) {
        int lk;
        const int zk = zone_e(k, Nz, a.np[1][0], a.np[1][1], lk);
        const bool row_in = k >= 1 && k < Nz, row_lt = k < Nz;
        const P hx = ldv<T, V>(Hx + idx), hz = ldv<T, V>(Hz + idx);
        const T hz_l = left_of<T, V>(hz, Hz + idx, i0, lane);
        P d = ldv<T, V>(a.Dy + idx);
        // +dHx/dz on columns 1..Nx-1, then -dHz/dx on rows 1..Nz-1 (FDTD2D.cpp:289-357)
        apply_uniform<+1, T, V>(d, diff<T, V>(hx, hx_km), t.m_in, zk, lk, face_z, 0, a.dtd[1], a.dt,
                                a.ph_ey_z[0] + (size_t)lk * px + i0, a.ph_ey_z[1] + (size_t)lk * px + i0);
        apply_x<-1, false, false, T, V>(d, diff_back_x<T, V>(hz, hz_l), row_in ? t.m_lt : 0u, i0, Nx, a.np[0][0], a.np[0][1], a.xrow, px,
                                 a.dtd[0], a.dt, t.lane_n ? a.ph_ey_x[0] + (size_t)k * a.xwn + i0 : nullptr,
                                 t.lane_p ? a.ph_ey_x[1] + (size_t)k * a.xwp + (i0 - a.xp0) : nullptr);
        if (a.src_on && k == a.kJ && a.iJ >= i0 && a.iJ < i0 + V) {
#pragma unroll
            for (int e = 0; e < V; ++e) if (i0 + e == a.iJ) d.v[e] = d.v[e] - a.jy;      // :359-362
        }
        stv<T, V>(a.Dy + idx, d);
        if (row_lt) {                                                                      // :365-379
            // A tile without dispersive cells (C1 = C2 = C3 = 0 throughout) reduces the recursion to
            // Ey = Dy/eps0 and S = 0: no history is read and the polarisation sums are left alone.
            const bool dispersive = __builtin_amdgcn_readfirstlane((int)a.disp[(size_t)k * a.ntx + t.tx]) != 0;
            if (dispersive) {
                const P c1 = ldv<T, V>(a.C1 + idx), c2 = ldv<T, V>(a.C2 + idx), c3 = ldv<T, V>(a.C3 + idx);
                const P f1 = ldv<T, V>(a.F1[0] + idx), s1 = ldv<T, V>(a.S1[0] + idx);
                P f2 = ldv<T, V>(a.F2[0] + idx), s2 = ldv<T, V>(a.S2[0] + idx);
#pragma unroll
                for (int e = 0; e < V; ++e)
                    if (t.m_lt & (1u << e)) {
                        T f, s;
                        lorentz<T>(d.v[e] / a.eps0, c1.v[e], c2.v[e], c3.v[e], f1.v[e], f2.v[e], s1.v[e], s2.v[e], f, s);
                        f2.v[e] = f; s2.v[e] = s;
                    }
                stv<T, V>(a.F2[0] + idx, f2);
                stv<T, V>(a.S2[0] + idx, s2);
            } else {
                P f2 = zero_pack<T, V>();
#pragma unroll
                for (int e = 0; e < V; ++e) if (t.m_lt & (1u << e)) f2.v[e] = d.v[e] / a.eps0;
                stv<T, V>(a.F2[0] + idx, f2);
            }
        }
        hx_km = hx;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) k2_part_h(const K2<T> a) {
    constexpr int V = VecWidth<T>::V;
    using P = Pack<T, V>;
    const Tile2<T> t = decode_tile2(a);
    if (!t.ok) return;
    const int i0 = t.i0, lane = t.lane, px = a.px, Nx = a.Nx, Nz = a.Nz;
    const T* __restrict__ Ey = a.F1[0];      // the sample the E part just produced
    const T* face_z = a.tab + (size_t)(1 * 2) * 6 * kMaxLayers;

    size_t idx = (size_t)t.kb * px + i0;
    P ey = ldv<T, V>(Ey + idx);
    for (int k = t.kb; k < t.ke; ++k, idx += px) {
        P ey_kp = zero_pack<T, V>();
        if (k + 1 < a.nz1) ey_kp = ldv<T, V>(Ey + idx + px);
        if (k < Nz) {
            int lk;
            const int zk = zone_h(k, Nz, a.np[1][0], a.np[1][1], false, lk);
            const bool row_in = k >= 1;
            const T ey_r = right_of<T, V>(ey, Ey + idx, i0, lane, px);
            const bool dispersive = __builtin_amdgcn_readfirstlane((int)a.disp[(size_t)k * a.ntx + t.tx]) != 0;
            P c1 = zero_pack<T, V>(), c2 = c1, c3 = c1;
            if (dispersive) { c1 = ldv<T, V>(a.C1 + idx); c2 = ldv<T, V>(a.C2 + idx); c3 = ldv<T, V>(a.C3 + idx); }
            // ---- Bx += dEy/dz on columns 1..Nx-1 (FDTD2D.cpp:383-416), then Hx (:460-468)
            {
                P b = ldv<T, V>(a.Bx + idx);
                apply_uniform<+1, T, V>(b, diff<T, V>(ey_kp, ey), t.m_in, zk, lk, face_z, 3, a.dtd[1], a.dt,
                                        a.ph_mx_z[0] + (size_t)lk * px + i0, a.ph_mx_z[1] + (size_t)lk * px + i0);
                stv<T, V>(a.Bx + idx, b);
                if (dispersive) {
                    const P f1 = ldv<T, V>(a.F1[1] + idx), s1 = ldv<T, V>(a.S1[1] + idx);
                    P f2 = ldv<T, V>(a.F2[1] + idx), s2 = ldv<T, V>(a.S2[1] + idx);
#pragma unroll
                    for (int e = 0; e < V; ++e)
                        if (t.m_lt & (1u << e)) {
                            T f, s;
                            lorentz<T>(b.v[e] / a.mu0, c1.v[e], c2.v[e], c3.v[e], f1.v[e], f2.v[e], s1.v[e], s2.v[e], f, s);
                            f2.v[e] = f; s2.v[e] = s;
                        }
                    stv<T, V>(a.F2[1] + idx, f2);
                    stv<T, V>(a.S2[1] + idx, s2);
                } else {
                    P f2 = zero_pack<T, V>();
#pragma unroll
                    for (int e = 0; e < V; ++e) if (t.m_lt & (1u << e)) f2.v[e] = b.v[e] / a.mu0;
                    stv<T, V>(a.F2[1] + idx, f2);
                }
            }
            // ---- Bz -= dEy/dx on rows 1..Nz-1 (FDTD2D.cpp:420-453), then Hz (:470-477)
            {
                P b = ldv<T, V>(a.Bz + idx);
                apply_x<-1, true, false, T, V>(b, diff_fwd_x<T, V>(ey, ey_r), row_in ? t.m_lt : 0u, i0, Nx, a.np[0][0], a.np[0][1], a.xrow, px,
                                        a.dtd[0], a.dt, t.lane_n ? a.ph_mz_x[0] + (size_t)k * a.xwn + i0 : nullptr,
                                        t.lane_p ? a.ph_mz_x[1] + (size_t)k * a.xwp + (i0 - a.xp0) : nullptr);
                stv<T, V>(a.Bz + idx, b);
                if (dispersive) {
                    const P f1 = ldv<T, V>(a.F1[2] + idx), s1 = ldv<T, V>(a.S1[2] + idx);
                    P f2 = ldv<T, V>(a.F2[2] + idx), s2 = ldv<T, V>(a.S2[2] + idx);
#pragma unroll
                    for (int e = 0; e < V; ++e)
                        if (t.m_lt & (1u << e)) {
                            T f, s;
                            lorentz<T>(b.v[e] / a.mu0, c1.v[e], c2.v[e], c3.v[e], f1.v[e], f2.v[e], s1.v[e], s2.v[e], f, s);
                            f2.v[e] = f; s2.v[e] = s;
                        }
                    stv<T, V>(a.F2[2] + idx, f2);
                    stv<T, V>(a.S2[2] + idx, s2);
                } else {
                    P f2 = zero_pack<T, V>();
#pragma unroll
                    for (int e = 0; e < V; ++e) if (t.m_lt & (1u << e)) f2.v[e] = b.v[e] / a.mu0;
                    stv<T, V>(a.F2[2] + idx, f2);
                }
            }
        }
        ey = ey_kp;
    }
}

}  // namespace fdtd
