// yee_twin.hpp -- CPU restatement ("twin") of the reference FDTD hot path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing the product ships links, imports or calls this file;
// it is the checker the HIP engine is compared against (tests/, the smoke check and the
// cpu_baseline leg of bench.py).
//
// What it restates (all paths relative to the reference checkout):
//   3-D  FDTD3D::updateE / applySource / updateElectricMaterialRelations
//        FDTD3D::updateH / updateMagneticMaterialRelations     fdtd3d/src/FDTD3D.cpp:480-1021
//        setupSource / setupMaterials / setupPML                fdtd3d/src/FDTD3D.cpp:69-463
//   2-D  FDTD2D::stepOnce                                       fdtd2d/src/FDTD2D.cpp:275-482
//        setupSource / setupMaterialsLorentz / setupPML         fdtd2d/src/FDTD2D.cpp:58-273
//
// The reference walks each array in 18 sweeps per half step (PML-, PML+, middle, per axis).
// Every cell receives at most one contribution per axis, each computed from the *other*
// field family only, so the result of a half step is independent of sweep order; what
// matters for bit-exactness is the order of the floating-point operations per cell.  The
// twin therefore visits each cell once and applies, in the reference's order, first-axis
// term, second-axis term, source, constitutive division.  Pinning: tests/test_oracle_ref.py
// compares it bit-for-bit against the compiled reference (oracle/_ref) where that is
// available, and tests/golden/ holds vectors generated from the compiled reference.
//
// Templated on the arithmetic type so the same code is the fp64 twin (== reference) and the
// fp32 twin (what the fp32 GPU path must match bit-for-bit).  Compile with
// -ffp-contract=off and without -ffast-math.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace twin {

// ---------------------------------------------------------------- helpers --
struct Box3 {
    int n[3];
    size_t size() const { return (size_t)n[0] * n[1] * n[2]; }
    size_t at(int i, int j, int k) const { return ((size_t)i * n[1] + j) * n[2] + k; }
    size_t at(const int* p) const { return at(p[0], p[1], p[2]); }
};
// D/E/eps component c: N along its own axis, N+1 along the others (FDTD3D.h:31-44).
inline Box3 boxE(int c, const int N[3]) { Box3 b; for (int a = 0; a < 3; ++a) b.n[a] = N[a] + (a != c); return b; }
// B/H/mu component c: N+1 along its own axis, N along the others (FDTD3D.h:46-51).
inline Box3 boxH(int c, const int N[3]) { Box3 b; for (int a = 0; a < 3; ++a) b.n[a] = N[a] + (a == c); return b; }

// Counter-based uniform [0,1) used for synthetic materials; identical on host and device.
inline double hash_unit(uint64_t seed, uint32_t stream, uint64_t index) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (index + 1) + ((uint64_t)stream << 56);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

enum Zone { Z_NONE = 0, Z_NEG = 1, Z_POS = 2, Z_MID = 3 };
// Where index `idx` along an axis of N cells falls for a backward-difference (D) term:
// low CPML idx = 1+l, high CPML idx = N-np+l, middle [nn+1, N-np)  (FDTD3D.cpp:540-577).
inline Zone zoneE(int idx, int N, int nn, int np, int& l) {
    if (idx < 1 || idx >= N) return Z_NONE;
    if (idx <= nn) { l = idx - 1; return Z_NEG; }
    if (idx >= N - np) { l = idx - (N - np); return Z_POS; }
    return Z_MID;
}
// Forward-difference (B) term: low CPML idx = l, high CPML idx = N-np+l, middle
// [nn, min(N-1, N-np)) in 3-D (FDTD3D.cpp:794-829: the last cell is skipped when np == 0),
// [nn, min(N-1, N-np-1)] in 2-D (FDTD2D.cpp:383-416: it is not).
inline Zone zoneH(int idx, int N, int nn, int np, int& l, bool skip_last_if_open) {
    if (idx < 0 || idx >= N) return Z_NONE;
    if (idx < nn) { l = idx; return Z_NEG; }
    if (idx >= N - np) {
        if (np == 0) return Z_NONE;
        l = idx - (N - np); return Z_POS;
    }
    if (skip_last_if_open && np == 0 && idx == N - 1) return Z_NONE;
    return Z_MID;
}

// One CPML face, six tables of `n` entries (FDTD3D.cpp:226-257 and the five copies below it).
struct FaceTabs { std::vector<double> t[6]; };   // b_e a_e C_e b_m a_m C_m
// side 0: low face (rho runs down from the wall), 1: high face.  d_step is the cell size
// used for sigma_max and C; d_den the cell size in the denominators of a_e/a_m (they differ
// only on the 2-D x faces, FDTD2D.cpp:161-162,192-193).
inline FaceTabs cpml_face(int n, int side, double d_step, double d_den, double dt, double eps0, double mu0) {
    FaceTabs f;
    for (auto& v : f.t) v.assign(n, 0.0);
    const int m = 3; const double alpha_max = 0.05, kappa_max = 7.0;
    const double sigma_max = 1 * (m + 1) / (150.0 * M_PI * std::sqrt(1.0 * 1.0) * d_step);
    for (int j = 0; j < n; ++j) {
        double rho_e, rho_m;
        if (side == 0) { rho_e = ((n - 0.75) - j) / n; rho_m = ((n - 0.25) - j) / n; }
        else           { rho_e = (0.25 + j) / n;       rho_m = (0.75 + j) / n; }
        double sig_e = sigma_max * std::pow(rho_e, m);
        double sig_m = (mu0 / eps0) * sigma_max * std::pow(rho_m, m);
        double kap_e = 1.0 + (kappa_max - 1.0) * std::pow(rho_e, m);
        double kap_m = 1.0 + (kappa_max - 1.0) * std::pow(rho_m, m);
        double alp_e = alpha_max * (1.0 - rho_e);
        double alp_m = (mu0 / eps0) * alpha_max * (1.0 - rho_m);
        double be = std::exp(-(sig_e / kap_e + alp_e) * dt / eps0);
        double bm = std::exp(-(sig_m / kap_m + alp_m) * dt / mu0);
        double den_e = d_den * kap_e * (sig_e + alp_e * kap_e);
        double den_m = d_den * kap_m * (sig_m + alp_m * kap_m);
        f.t[0][j] = be;
        f.t[1][j] = (std::abs(den_e) < 1e-30) ? 0.0 : (sig_e * (be - 1.0) / den_e);
        f.t[2][j] = (dt / d_step) / kap_e;
        f.t[3][j] = bm;
        f.t[4][j] = (std::abs(den_m) < 1e-30) ? 0.0 : (sig_m * (bm - 1.0) / den_m);
        f.t[5][j] = (dt / d_step) / kap_m;
    }
    return f;
}

struct Consts {
    double c0 = 299792458.0;
    double mu0 = 4.0 * M_PI * 1e-7;
    double eps0 = 1.0 / (c0 * c0 * mu0);
    double Z0 = std::sqrt(mu0 / eps0);
};

// ------------------------------------------------------------------- 3-D --
struct Term { int axis, src, sign; };
// Order of the two curl terms per component (FDTD3D.cpp:538-778, :792-1018).
static const Term kTermE[3][2] = {{{2, 1, -1}, {1, 2, +1}},    // Dx: -dHy/dz, +dHz/dy
                                  {{2, 0, +1}, {0, 2, -1}},    // Dy: +dHx/dz, -dHz/dx
                                  {{0, 1, +1}, {1, 0, -1}}};   // Dz: +dHy/dx, -dHx/dy
static const Term kTermH[3][2] = {{{2, 1, +1}, {1, 2, -1}},    // Bx: +dEy/dz, -dEz/dy
                                  {{0, 2, +1}, {2, 0, -1}},    // By: +dEz/dx, -dEx/dz
                                  {{1, 0, +1}, {0, 1, -1}}};   // Bz: +dEx/dy, -dEy/dx

template <class T>
struct Twin3D {
    int N[3] = {0, 0, 0};
    int np[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    int kJ = 0;
    bool has_mu = false;
    T dt = 0, dtd[3] = {0, 0, 0}, mu0 = 0;
    std::vector<T> D[3], E[3], B[3], H[3], eps[3], mu[3];
    std::vector<T> tab[3][2][6];
    std::vector<T> phiE[3][3][2], phiM[3][3][2];   // [component][axis][side]
    std::vector<T> mag, wf;

    void init(int Nx, int Ny, int Nz, const int npml[6], double dx, double dy, double dz, double dt_,
              double mu0_, int kJ_, int n_wf, bool has_mu_) {
        N[0] = Nx; N[1] = Ny; N[2] = Nz;
        for (int a = 0; a < 3; ++a) for (int s = 0; s < 2; ++s) np[a][s] = npml[a * 2 + s];
        kJ = kJ_; has_mu = has_mu_;
        dt = (T)dt_; dtd[0] = (T)(dt_ / dx); dtd[1] = (T)(dt_ / dy); dtd[2] = (T)(dt_ / dz); mu0 = (T)mu0_;
        for (int c = 0; c < 3; ++c) {
            size_t ne = boxE(c, N).size(), nh = boxH(c, N).size();
            D[c].assign(ne, 0); E[c].assign(ne, 0); eps[c].assign(ne, 1);
            B[c].assign(nh, 0); H[c].assign(nh, 0); mu[c].assign(has_mu ? nh : 0, 1);
            for (int a = 0; a < 3; ++a) for (int s = 0; s < 2; ++s) {
                phiE[c][a][s].clear(); phiM[c][a][s].clear();
                if (a == c) continue;
                phiE[c][a][s].assign(phiBox(false, c, a, s).size(), 0);
                phiM[c][a][s].assign(phiBox(true, c, a, s).size(), 0);
            }
        }
        for (int a = 0; a < 3; ++a) for (int s = 0; s < 2; ++s) for (int q = 0; q < 6; ++q) tab[a][s][q].assign(np[a][s], 0);
        mag.assign((size_t)Nx * (Ny + 1), 0); wf.assign(n_wf, 0);
    }
    Box3 phiBox(bool magnetic, int c, int a, int s) const {
        Box3 b = magnetic ? boxH(c, N) : boxE(c, N);
        b.n[a] = np[a][s];
        return b;
    }

    // One CPML / middle contribution (FDTD3D.cpp:545-548 and :572-573 are the pattern).
    inline T contribution(Zone z, int l, T curl, int a, bool magnetic, std::vector<T>* phi, const Box3* pb, int* p) const {
        if (z == Z_MID) return dtd[a] * curl;
        int s = (z == Z_NEG) ? 0 : 1;
        const std::vector<T>* tb = tab[a][s];
        int save = p[a]; p[a] = l;
        T& ph = phi[s][pb[s].at(p)];
        p[a] = save;
        ph = tb[magnetic ? 3 : 0][l] * ph + tb[magnetic ? 4 : 1][l] * curl;
        return tb[magnetic ? 5 : 2][l] * curl + dt * ph;
    }

    void half_e(int n) {
        const bool src_on = n >= 0 && n < (int)wf.size();
        const T jx = src_on ? dt * wf[n] : (T)0;                        // FDTD3D.cpp:482
        for (int c = 0; c < 3; ++c) {
            const Box3 bc = boxE(c, N);
            Box3 bsrc[2], pb[2][2];
            for (int t = 0; t < 2; ++t) {
                bsrc[t] = boxH(kTermE[c][t].src, N);
                for (int s = 0; s < 2; ++s) pb[t][s] = phiBox(false, c, kTermE[c][t].axis, s);
            }
            int p[3];
            for (p[0] = 0; p[0] < bc.n[0]; ++p[0]) for (p[1] = 0; p[1] < bc.n[1]; ++p[1]) for (p[2] = 0; p[2] < bc.n[2]; ++p[2]) {
                const size_t id = bc.at(p);
                T d = D[c][id];
                for (int t = 0; t < 2; ++t) {
                    const Term& tm = kTermE[c][t];
                    const int a = tm.axis, o = 3 - c - a;       // o: the remaining axis
                    if (p[c] >= N[c]) continue;
                    // rows touched by the sweep: the Dx z-sweep starts at j = 0, every other
                    // sweep at 1 along the remaining axis (FDTD3D.cpp:541 vs :582-583 etc.)
                    const int lo = (c == 0 && a == 2) ? 0 : 1;
                    if (p[o] < lo || p[o] >= N[o]) continue;
                    int l = 0;
                    const Zone z = zoneE(p[a], N[a], np[a][0], np[a][1], l);
                    if (z == Z_NONE) continue;
                    const std::vector<T>& h = H[tm.src];
                    int q[3] = {p[0], p[1], p[2]};
                    const T hi = h[bsrc[t].at(q)];
                    q[a] -= 1;
                    const T curl = hi - h[bsrc[t].at(q)];
                    const T v = contribution(z, l, curl, a, false, phiE[c][a], pb[t], p);
                    d = (tm.sign > 0) ? d + v : d - v;
                }
                if (c == 0 && src_on && p[2] == kJ) d = d + jx * mag[(size_t)p[0] * (N[1] + 1) + p[1]];   // :485-490
                D[c][id] = d;
                E[c][id] = d / eps[c][id];                                                                  // :494-521
            }
        }
    }

    void half_h() {
        for (int c = 0; c < 3; ++c) {
            const Box3 bc = boxH(c, N);
            Box3 bsrc[2], pb[2][2];
            for (int t = 0; t < 2; ++t) {
                bsrc[t] = boxE(kTermH[c][t].src, N);
                for (int s = 0; s < 2; ++s) pb[t][s] = phiBox(true, c, kTermH[c][t].axis, s);
            }
            int p[3];
            for (p[0] = 0; p[0] < bc.n[0]; ++p[0]) for (p[1] = 0; p[1] < bc.n[1]; ++p[1]) for (p[2] = 0; p[2] < bc.n[2]; ++p[2]) {
                const size_t id = bc.at(p);
                T b = B[c][id];
                for (int t = 0; t < 2; ++t) {
                    const Term& tm = kTermH[c][t];
                    const int a = tm.axis, o = 3 - c - a;
                    if (p[c] < 1 || p[c] >= N[c]) continue;      // FDTD3D.cpp:794, :870, :947
                    if (p[o] >= N[o]) continue;
                    int l = 0;
                    const Zone z = zoneH(p[a], N[a], np[a][0], np[a][1], l, true);
                    if (z == Z_NONE) continue;
                    const std::vector<T>& e = E[tm.src];
                    int q[3] = {p[0], p[1], p[2]};
                    const T lo = e[bsrc[t].at(q)];
                    q[a] += 1;
                    const T curl = e[bsrc[t].at(q)] - lo;
                    const T v = contribution(z, l, curl, a, true, phiM[c][a], pb[t], p);
                    b = (tm.sign > 0) ? b + v : b - v;
                }
                B[c][id] = b;
                H[c][id] = has_mu ? b / mu[c][id] : b / mu0;                                               // :523-527
            }
        }
    }
    void step(int n) { half_e(n); half_h(); }

    // Cell-centre observables of FDTD3D::run / dumpFrame (FDTD3D.cpp:1023-1097); arrays [Nz][Ny][Nx],
    // x fastest.  Sums run in the reference's order; the square root is taken in double.
    std::vector<T> acc;
    void edge_sums(int i, int j, int k, T& sx, T& sy, T& sz) const {
        const Box3 bx = boxE(0, N), by = boxE(1, N), bz = boxE(2, N);
        sx = E[0][bx.at(i, j, k)] + E[0][bx.at(i, j + 1, k)] + E[0][bx.at(i, j, k + 1)] + E[0][bx.at(i, j + 1, k + 1)];
        sy = E[1][by.at(i, j, k)] + E[1][by.at(i + 1, j, k)] + E[1][by.at(i, j, k + 1)] + E[1][by.at(i + 1, j, k + 1)];
        sz = E[2][bz.at(i, j, k)] + E[2][bz.at(i + 1, j, k)] + E[2][bz.at(i, j + 1, k)] + E[2][bz.at(i + 1, j + 1, k)];
    }
    void observe_accumulate(double weight, bool reset) {
        acc.resize((size_t)N[0] * N[1] * N[2], 0);
        if (reset) std::fill(acc.begin(), acc.end(), (T)0);
        for (int k = 0; k < N[2]; ++k) for (int j = 0; j < N[1]; ++j) for (int i = 0; i < N[0]; ++i) {
            T sx, sy, sz;
            edge_sums(i, j, k, sx, sy, sz);
            const T e_sq = (T)0.0625 * (sx * sx + sy * sy + sz * sz);
            T& a = acc[((size_t)k * N[1] + j) * N[0] + i];
            a = a + e_sq * (T)weight;
        }
    }
    void observe_rms(double* out) const { for (size_t q = 0; q < acc.size(); ++q) out[q] = std::sqrt((double)acc[q]); }
    void observe_frame(double* ex_center, double* e_mag) const {
        for (int k = 0; k < N[2]; ++k) for (int j = 0; j < N[1]; ++j) for (int i = 0; i < N[0]; ++i) {
            T sx, sy, sz;
            edge_sums(i, j, k, sx, sy, sz);
            const T ax = (T)0.25 * sx, ay = (T)0.25 * sy, az = (T)0.25 * sz;
            const size_t q = ((size_t)k * N[1] + j) * N[0] + i;
            ex_center[q] = (double)ax;
            e_mag[q] = std::sqrt((double)(T)(ax * ax + ay * ay + az * az));
        }
    }

    // eps_c = lo + (hi-lo)*u(seed, c, canonical index); mu likewise with stream 3+c.
    void fill_materials(uint64_t seed, double eps_lo, double eps_hi, double mu_lo, double mu_hi) {
        for (int c = 0; c < 3; ++c) {
            for (size_t id = 0; id < eps[c].size(); ++id) eps[c][id] = (T)(eps_lo + (eps_hi - eps_lo) * hash_unit(seed, c, id));
            if (has_mu) for (size_t id = 0; id < mu[c].size(); ++id) mu[c][id] = (T)(mu_lo + (mu_hi - mu_lo) * hash_unit(seed, 3 + c, id));
        }
    }
};

// Reference-default setup pieces, fp64 (FDTD3D.cpp:69-200).
inline void default_source3d(int Nx, int Ny, int n_wf, double dx, double dy, double dz, double dt,
                             std::vector<double>& wf, std::vector<double>& mag) {
    Consts K;
    const double lambda0 = 1053e-9, E0 = 1.0;
    const double f0 = K.c0 / lambda0, omega0 = 2.0 * M_PI * f0, T0 = 1.0 / f0;
    wf.assign(n_wf, 0.0);
    for (int n = 0; n < n_wf; ++n) {
        double t = (n + 0.5) * dt;
        wf[n] = (2.0 * E0 / K.Z0) * std::sin(omega0 * t) / dz;
    }
    const double ts = 5.0 * T0;
    const int tsddt = (int)std::round(ts / dt);
    for (int n = 0; n <= tsddt && n < n_wf; ++n) {
        double s = (double)n / tsddt;
        double ws = 10.0 * std::pow(s, 3) - 15.0 * std::pow(s, 4) + 6.0 * std::pow(s, 5);
        wf[n] *= ws;
    }
    mag.assign((size_t)Nx * (Ny + 1), 0.0);
    const double w0 = 1.5 * lambda0;
    const int xc = Nx / 2, yc = Ny / 2;
    for (int i = 0; i < Nx; ++i) for (int j = 0; j <= Ny; ++j) {
        double x = (i - xc) * dx, y = (j - yc) * dy;
        mag[(size_t)i * (Ny + 1) + j] = E0 * std::exp(-(x * x + y * y) / (w0 * w0));
    }
}

// Fused-silica background with a centred air sphere, 3x3 sub-sampled on the two transverse
// axes of each component (FDTD3D.cpp:106-200).
inline void default_materials3d(const int N[3], double dx, double dy, double dz, std::vector<double> eps[3]) {
    Consts K;
    const double d[3] = {dx, dy, dz};
    const double eps_bg = (1.47 * 1.47) * K.eps0, eps_in = (1.0 * 1.0) * K.eps0, R = 2.0 * 1053e-9;
    const int ctr[3] = {N[0] / 2, N[1] / 2, N[2] / 2};
    for (int c = 0; c < 3; ++c) {
        const Box3 b = boxE(c, N);
        eps[c].assign(b.size(), eps_bg);
        const int u = (c == 0) ? 1 : 0, v = (c == 2) ? 1 : 2;      // the two sub-sampled axes, in loop order
        int p[3];
        for (p[0] = 0; p[0] < b.n[0]; ++p[0]) for (p[1] = 0; p[1] < b.n[1]; ++p[1]) for (p[2] = 0; p[2] < b.n[2]; ++p[2]) {
            double pos[3];
            for (int a = 0; a < 3; ++a) pos[a] = (a == c) ? ((p[a] + 0.5) - ctr[a]) * d[a] : (p[a] - ctr[a]) * d[a];
            int inside = 0;
            for (int uu = -1; uu <= 1; ++uu) {
                double pu = pos[u] + uu * (d[u] * 0.5);
                for (int vv = -1; vv <= 1; ++vv) {
                    double pv = pos[v] + vv * (d[v] * 0.5);
                    double xyz[3]; xyz[c] = pos[c]; xyz[u] = pu; xyz[v] = pv;
                    double r = std::sqrt(xyz[0] * xyz[0] + xyz[1] * xyz[1] + xyz[2] * xyz[2]);
                    if (r <= R) ++inside;
                }
            }
            double fraction = (double)inside / 9.0;
            eps[c][b.at(p)] = eps_bg + (eps_in - eps_bg) * fraction;
        }
    }
}

// ------------------------------------------------------------------- 2-D --
template <class T>
struct Twin2D {
    int Nx = 0, Nz = 0, np[2][2] = {{0, 0}, {0, 0}};   // axis 0 = x, 1 = z
    int iJ = 0, kJ = 0;
    T dt = 0, dtd[2] = {0, 0}, mu0 = 0, eps0 = 0;
    // field families: 0 = Ey-like [Nx+1][Nz+1], 1 = Hx-like [Nx+1][Nz], 2 = Hz-like [Nx][Nz+1]
    std::vector<T> Dy, Bx, Bz;
    std::vector<T> F[3], F1[3], F2[3], S[3], S1[3], S2[3];   // field, its two histories, polarisation sums
    std::vector<T> C1, C2, C3;
    std::vector<T> tab[2][2][6];
    std::vector<T> phi_ey[2][2], phi_mx_z[2], phi_mz_x[2];   // phi_ey[axis][side]
    std::vector<T> Jy;

    size_t idE(int i, int k) const { return (size_t)i * (Nz + 1) + k; }
    size_t idHx(int i, int k) const { return (size_t)i * Nz + k; }
    size_t idHz(int i, int k) const { return (size_t)i * (Nz + 1) + k; }
    size_t idC(int i, int k) const { return (size_t)i * Nz + k; }

    void init(int Nx_, int Nz_, const int npml[4], double dx, double dz, double dt_, double mu0_, double eps0_,
              int iJ_, int kJ_, int n_wf) {
        Nx = Nx_; Nz = Nz_; iJ = iJ_; kJ = kJ_;
        for (int a = 0; a < 2; ++a) for (int s = 0; s < 2; ++s) np[a][s] = npml[a * 2 + s];
        dt = (T)dt_; dtd[0] = (T)(dt_ / dx); dtd[1] = (T)(dt_ / dz); mu0 = (T)mu0_; eps0 = (T)eps0_;
        const size_t n[3] = {(size_t)(Nx + 1) * (Nz + 1), (size_t)(Nx + 1) * Nz, (size_t)Nx * (Nz + 1)};
        Dy.assign(n[0], 0); Bx.assign(n[1], 0); Bz.assign(n[2], 0);
        for (int f = 0; f < 3; ++f) { F[f].assign(n[f], 0); F1[f] = F2[f] = S[f] = S1[f] = S2[f] = F[f]; }
        C1.assign((size_t)Nx * Nz, 0); C2 = C3 = C1;
        for (int a = 0; a < 2; ++a) for (int s = 0; s < 2; ++s) for (int q = 0; q < 6; ++q) tab[a][s][q].assign(np[a][s], 0);
        for (int s = 0; s < 2; ++s) {
            phi_ey[0][s].assign((size_t)np[0][s] * (Nz + 1), 0);       // [l][k]
            phi_ey[1][s].assign((size_t)(Nx + 1) * np[1][s], 0);       // [i][l]
            phi_mx_z[s].assign((size_t)(Nx + 1) * np[1][s], 0);        // [i][l]
            phi_mz_x[s].assign((size_t)np[0][s] * (Nz + 1), 0);        // [l][k]
        }
        Jy.assign(n_wf, 0);
    }

    inline T contribution(Zone z, int l, T curl, int a, bool magnetic, T& ph_neg, T& ph_pos) const {
        if (z == Z_MID) return dtd[a] * curl;
        const int s = (z == Z_NEG) ? 0 : 1;
        T& ph = s ? ph_pos : ph_neg;
        const std::vector<T>* tb = tab[a][s];
        ph = tb[magnetic ? 3 : 0][l] * ph + tb[magnetic ? 4 : 1][l] * curl;
        return tb[magnetic ? 5 : 2][l] * curl + dt * ph;
    }

    // Second-order Lorentz recursion for one sample (FDTD2D.cpp:369-373, :463-476).
    inline void lorentz(T drive, T c1, T c2, T c3, T f1, T f2, T s1, T s2, T& f, T& s) const {
        const T num = drive - (T)2.0 * c1 * f1 - c1 * f2 + c2 * s1 + c3 * s2;
        f = num / ((T)1.0 + c1);
        s = c1 * f + (T)2.0 * c1 * f1 + c1 * f2 - c2 * s1 - c3 * s2;
    }
    void rotate(int f) {     // FDTD2D.cpp:376-379, :480-481
        F2[f].swap(F1[f]); F1[f] = F[f];
        S2[f].swap(S1[f]); S1[f] = S[f];
    }

    // RMS window of Ey (FDTD2D.cpp:515-522), [Nx][Nz] with k fastest
    std::vector<T> acc;
    void observe_accumulate(double weight, bool reset) {
        acc.resize((size_t)Nx * Nz, 0);
        if (reset) std::fill(acc.begin(), acc.end(), (T)0);
        for (int i = 0; i < Nx; ++i) for (int k = 0; k < Nz; ++k) {
            const T e = F[0][idE(i, k)];
            T& a = acc[(size_t)i * Nz + k];
            a = a + e * e * (T)weight;
        }
    }
    void observe_rms(double* out) const { for (size_t q = 0; q < acc.size(); ++q) out[q] = std::sqrt((double)acc[q]); }

    void step(int n) {
        T dummy = 0;
        // ---- Dy: +dHx/dz then -dHz/dx, then the line source (FDTD2D.cpp:289-362)
        for (int i = 0; i <= Nx; ++i) for (int k = 0; k <= Nz; ++k) {
            T d = Dy[idE(i, k)];
            int l = 0;
            if (i >= 1 && i <= Nx - 1) {
                Zone z = zoneE(k, Nz, np[1][0], np[1][1], l);
                if (z != Z_NONE) {
                    T curl = F[1][idHx(i, k)] - F[1][idHx(i, k - 1)];
                    T& pn = (z == Z_NEG) ? phi_ey[1][0][(size_t)i * np[1][0] + l] : dummy;
                    T& pp = (z == Z_POS) ? phi_ey[1][1][(size_t)i * np[1][1] + l] : dummy;
                    d = d + contribution(z, l, curl, 1, false, pn, pp);
                }
            }
            if (k >= 1 && k <= Nz - 1) {
                Zone z = zoneE(i, Nx, np[0][0], np[0][1], l);
                if (z != Z_NONE) {
                    T curl = F[2][idHz(i, k)] - F[2][idHz(i - 1, k)];
                    T& pn = (z == Z_NEG) ? phi_ey[0][0][(size_t)l * (Nz + 1) + k] : dummy;
                    T& pp = (z == Z_POS) ? phi_ey[0][1][(size_t)l * (Nz + 1) + k] : dummy;
                    d = d - contribution(z, l, curl, 0, false, pn, pp);
                }
            }
            if (i == iJ && k == kJ && n >= 0 && n < (int)Jy.size()) d = d - dt * Jy[n];
            Dy[idE(i, k)] = d;
        }
        // ---- Ey (FDTD2D.cpp:365-379)
        for (int i = 0; i < Nx; ++i) for (int k = 0; k < Nz; ++k) {
            const size_t c = idC(i, k), e = idE(i, k);
            lorentz(Dy[e] / eps0, C1[c], C2[c], C3[c], F1[0][e], F2[0][e], S1[0][e], S2[0][e], F[0][e], S[0][e]);
        }
        rotate(0);
        // ---- Bx: +dEy/dz (FDTD2D.cpp:383-416)
        for (int i = 1; i <= Nx - 1; ++i) for (int k = 0; k < Nz; ++k) {
            int l = 0;
            Zone z = zoneH(k, Nz, np[1][0], np[1][1], l, false);
            if (z == Z_NONE) continue;
            T curl = F[0][idE(i, k + 1)] - F[0][idE(i, k)];
            T& pn = (z == Z_NEG) ? phi_mx_z[0][(size_t)i * np[1][0] + l] : dummy;
            T& pp = (z == Z_POS) ? phi_mx_z[1][(size_t)i * np[1][1] + l] : dummy;
            Bx[idHx(i, k)] = Bx[idHx(i, k)] + contribution(z, l, curl, 1, true, pn, pp);
        }
        // ---- Bz: -dEy/dx (FDTD2D.cpp:420-453)
        for (int i = 0; i < Nx; ++i) for (int k = 1; k <= Nz - 1; ++k) {
            int l = 0;
            Zone z = zoneH(i, Nx, np[0][0], np[0][1], l, false);
            if (z == Z_NONE) continue;
            T curl = F[0][idE(i + 1, k)] - F[0][idE(i, k)];
            T& pn = (z == Z_NEG) ? phi_mz_x[0][(size_t)l * (Nz + 1) + k] : dummy;
            T& pp = (z == Z_POS) ? phi_mz_x[1][(size_t)l * (Nz + 1) + k] : dummy;
            Bz[idHz(i, k)] = Bz[idHz(i, k)] - contribution(z, l, curl, 0, true, pn, pp);
        }
        // ---- Hx, Hz (FDTD2D.cpp:456-481)
        for (int i = 0; i < Nx; ++i) for (int k = 0; k < Nz; ++k) {
            const size_t c = idC(i, k);
            { const size_t h = idHx(i, k);
              lorentz(Bx[h] / mu0, C1[c], C2[c], C3[c], F1[1][h], F2[1][h], S1[1][h], S2[1][h], F[1][h], S[1][h]); }
            { const size_t h = idHz(i, k);
              lorentz(Bz[h] / mu0, C1[c], C2[c], C3[c], F1[2][h], F2[2][h], S1[2][h], S2[2][h], F[2][h], S[2][h]); }
        }
        rotate(1); rotate(2);
    }
};

// Reference-default 2-D setup pieces, fp64 (FDTD2D.cpp:58-112).
inline void default_source2d(int n_wf, double dx, double dz, double dt, std::vector<double>& Jy) {
    Consts K;
    const double f0 = 1.5e14, omega0 = 2.0 * M_PI * f0, T0 = 1.0 / f0, Jy0 = 1.0;
    Jy.assign(n_wf, 0.0);
    for (int n = 0; n < n_wf; ++n) {
        double t = (n + 1) * dt;
        Jy[n] = Jy0 * std::sin(omega0 * t) / (dx * dz) / (omega0 * K.mu0 / 4.0);
    }
    const double ts = 2.0 * T0;
    const int tsddt = (int)std::round(ts / dt);
    for (int n = 0; n <= tsddt && n < n_wf; ++n) {
        double s = (double)n / tsddt;
        double ws = s - std::sin(2.0 * M_PI * s) / (2.0 * M_PI);
        Jy[n] *= ws;
    }
}
// Lorentz slab ix in [Nx/2-20, Nx/2+20], every kz (FDTD2D.cpp:84-112).
inline void default_lorentz2d(int Nx, int Nz, double dt, std::vector<double>& C1, std::vector<double>& C2, std::vector<double>& C3) {
    const double f0 = 1.5e14, omega0 = 2.0 * M_PI * f0, wp = omega0, w0 = wp / std::sqrt(2.0), delta0 = 0.0;
    const double wpb = wp * dt, w0b = w0 * dt, d0b = delta0 * dt;
    C1.assign((size_t)Nx * Nz, 0.0); C2 = C3 = C1;
    const int xc = Nx / 2, xl = xc - 40 / 2, xr = xc + 40 / 2;
    for (int ix = std::max(xl, 0); ix <= xr && ix < Nx; ++ix) for (int kz = 0; kz <= Nz - 1; ++kz) {
        const size_t c = (size_t)ix * Nz + kz;
        const double den = (w0b * w0b + 4.0 * d0b + 4.0);
        C1[c] = (wpb * wpb) / den;
        C2[c] = 2.0 * (w0b * w0b - 4.0) / den;
        C3[c] = (w0b * w0b - 4.0 * d0b + 4.0) / den;
    }
}

}  // namespace twin
