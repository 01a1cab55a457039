// yee_twin_capi.cpp -- C entry points of the CPU twin, for ctypes.  TEST INFRASTRUCTURE ONLY
// (see yee_twin.hpp).  Array ids and canonical host layouts are those of
// include/fdtd_engine.h, so a test can move the same numpy arrays between the twin, the
// compiled reference and the HIP engine.
#include "yee_twin.hpp"
#include "../include/fdtd_engine.h"

#include <cstring>
#include <memory>

using namespace twin;

namespace {

struct Any3D {
    int prec;
    Twin3D<float> f;
    Twin3D<double> d;
};
struct Any2D {
    int prec;
    Twin2D<float> f;
    Twin2D<double> d;
};

template <class T> std::vector<T>* find3(Twin3D<T>& t, int id) {
    if (id >= 0 && id < 18) {
        int c = id % 3;
        switch (id / 3) {
            case 0: return &t.D[c]; case 1: return &t.E[c]; case 2: return &t.B[c];
            case 3: return &t.H[c]; case 4: return &t.eps[c]; case 5: return t.has_mu ? &t.mu[c] : nullptr;
        }
    }
    if ((id >= FDTD3D_PHI_E && id < FDTD3D_PHI_E + 18) || (id >= FDTD3D_PHI_M && id < FDTD3D_PHI_M + 18)) {
        bool m = id >= FDTD3D_PHI_M;
        int r = id - (m ? FDTD3D_PHI_M : FDTD3D_PHI_E);
        int s = r % 2, a = (r / 2) % 3, c = r / 6;
        if (a == c) return nullptr;
        return m ? &t.phiM[c][a][s] : &t.phiE[c][a][s];
    }
    if (id >= FDTD3D_TAB && id < FDTD3D_TAB + 36) {
        int r = id - FDTD3D_TAB;
        return &t.tab[r / 12][(r / 6) % 2][r % 6];
    }
    if (id == FDTD3D_JX_MAG) return &t.mag;
    if (id == FDTD3D_JX_WF) return &t.wf;
    return nullptr;
}

template <class T> std::vector<T>* find2(Twin2D<T>& t, int id) {
    if (id >= 0 && id <= 20) {
        if (id == FDTD2D_DY) return &t.Dy;
        if (id == FDTD2D_BX) return &t.Bx;
        if (id == FDTD2D_BZ) return &t.Bz;
        int f = id / 7, r = id % 7;    // r: 1 F, 2 F1, 3 F2, 4 S, 5 S1, 6 S2
        switch (r) {
            case 1: return &t.F[f]; case 2: return &t.F1[f]; case 3: return &t.F2[f];
            case 4: return &t.S[f]; case 5: return &t.S1[f]; case 6: return &t.S2[f];
        }
    }
    if (id == FDTD2D_C1) return &t.C1;
    if (id == FDTD2D_C2) return &t.C2;
    if (id == FDTD2D_C3) return &t.C3;
    if (id >= FDTD2D_PHI && id < FDTD2D_PHI + 8) {
        int r = id - FDTD2D_PHI, s = r % 2;
        switch (r / 2) {
            case 0: return &t.phi_ey[1][s]; case 1: return &t.phi_ey[0][s];
            case 2: return &t.phi_mx_z[s];  case 3: return &t.phi_mz_x[s];
        }
    }
    if (id >= FDTD2D_TAB && id < FDTD2D_TAB + 24) {
        int r = id - FDTD2D_TAB;
        return &t.tab[r / 12][(r / 6) % 2][r % 6];
    }
    if (id == FDTD2D_JY) return &t.Jy;
    return nullptr;
}

template <class T> int put(std::vector<T>* v, const double* src, size_t n) {
    if (!v || v->size() != n) return 1;
    for (size_t i = 0; i < n; ++i) (*v)[i] = (T)src[i];
    return 0;
}
template <class T> int take(const std::vector<T>* v, double* dst, size_t n) {
    if (!v || v->size() != n) return 1;
    for (size_t i = 0; i < n; ++i) dst[i] = (double)(*v)[i];
    return 0;
}

}  // namespace

extern "C" {

void* twin3d_create(int prec, int Nx, int Ny, int Nz, const int npml[6], double dx, double dy, double dz,
                    double dt, double mu0, int kJ, int n_wf, int has_mu) {
    if (prec != 32 && prec != 64) return nullptr;
    const int N[3] = {Nx, Ny, Nz};
    for (int a = 0; a < 3; ++a) if (N[a] < 1 || npml[2 * a] < 0 || npml[2 * a + 1] < 0 || npml[2 * a] + npml[2 * a + 1] >= N[a]) return nullptr;
    auto* p = new Any3D;
    p->prec = prec;
    if (prec == 32) p->f.init(Nx, Ny, Nz, npml, dx, dy, dz, dt, mu0, kJ, n_wf, has_mu != 0);
    else p->d.init(Nx, Ny, Nz, npml, dx, dy, dz, dt, mu0, kJ, n_wf, has_mu != 0);
    return p;
}
void twin3d_destroy(void* h) { delete (Any3D*)h; }
size_t twin3d_size(void* h, int id) {
    auto* p = (Any3D*)h;
    if (p->prec == 32) { auto* v = find3(p->f, id); return v ? v->size() : 0; }
    auto* v = find3(p->d, id); return v ? v->size() : 0;
}
int twin3d_set(void* h, int id, const double* src, size_t n) {
    auto* p = (Any3D*)h;
    return p->prec == 32 ? put(find3(p->f, id), src, n) : put(find3(p->d, id), src, n);
}
int twin3d_get(void* h, int id, double* dst, size_t n) {
    auto* p = (Any3D*)h;
    return p->prec == 32 ? take(find3(p->f, id), dst, n) : take(find3(p->d, id), dst, n);
}
void twin3d_step(void* h, int n0, int count) {
    auto* p = (Any3D*)h;
    for (int n = n0; n < n0 + count; ++n) { if (p->prec == 32) p->f.step(n); else p->d.step(n); }
}
void twin3d_half_e(void* h, int n) { auto* p = (Any3D*)h; if (p->prec == 32) p->f.half_e(n); else p->d.half_e(n); }
void twin3d_half_h(void* h) { auto* p = (Any3D*)h; if (p->prec == 32) p->f.half_h(); else p->d.half_h(); }
void twin3d_fill_materials(void* h, uint64_t seed, double eps_lo, double eps_hi, double mu_lo, double mu_hi) {
    auto* p = (Any3D*)h;
    if (p->prec == 32) p->f.fill_materials(seed, eps_lo, eps_hi, mu_lo, mu_hi);
    else p->d.fill_materials(seed, eps_lo, eps_hi, mu_lo, mu_hi);
}

void twin3d_observe_accumulate(void* h, double weight, int reset) { auto* p = (Any3D*)h; if (p->prec == 32) p->f.observe_accumulate(weight, reset != 0); else p->d.observe_accumulate(weight, reset != 0); }
void twin3d_observe_rms(void* h, double* out) { auto* p = (Any3D*)h; if (p->prec == 32) p->f.observe_rms(out); else p->d.observe_rms(out); }
void twin3d_observe_frame(void* h, double* ex_center, double* e_mag) { auto* p = (Any3D*)h; if (p->prec == 32) p->f.observe_frame(ex_center, e_mag); else p->d.observe_frame(ex_center, e_mag); }

// fp64 setup generators (outputs sized by the caller).
void twin_consts(double out[4]) { Consts K; out[0] = K.c0; out[1] = K.mu0; out[2] = K.eps0; out[3] = K.Z0; }
void twin_cpml_face(int n, int side, double d_step, double d_den, double dt, double out[/*6*n*/]) {
    Consts K;
    FaceTabs f = cpml_face(n, side, d_step, d_den, dt, K.eps0, K.mu0);
    for (int q = 0; q < 6; ++q) std::memcpy(out + (size_t)q * n, f.t[q].data(), sizeof(double) * n);
}
void twin_default_source3d(int Nx, int Ny, int n_wf, double dx, double dy, double dz, double dt, double* wf, double* mag) {
    std::vector<double> w, m;
    default_source3d(Nx, Ny, n_wf, dx, dy, dz, dt, w, m);
    std::memcpy(wf, w.data(), sizeof(double) * w.size());
    std::memcpy(mag, m.data(), sizeof(double) * m.size());
}
void twin_default_materials3d(int Nx, int Ny, int Nz, double dx, double dy, double dz, double* ex, double* ey, double* ez) {
    const int N[3] = {Nx, Ny, Nz};
    std::vector<double> e[3];
    default_materials3d(N, dx, dy, dz, e);
    double* out[3] = {ex, ey, ez};
    for (int c = 0; c < 3; ++c) std::memcpy(out[c], e[c].data(), sizeof(double) * e[c].size());
}
void twin_default_source2d(int n_wf, double dx, double dz, double dt, double* Jy) {
    std::vector<double> j;
    default_source2d(n_wf, dx, dz, dt, j);
    std::memcpy(Jy, j.data(), sizeof(double) * j.size());
}
void twin_default_lorentz2d(int Nx, int Nz, double dt, double* C1, double* C2, double* C3) {
    std::vector<double> a, b, c;
    default_lorentz2d(Nx, Nz, dt, a, b, c);
    std::memcpy(C1, a.data(), sizeof(double) * a.size());
    std::memcpy(C2, b.data(), sizeof(double) * b.size());
    std::memcpy(C3, c.data(), sizeof(double) * c.size());
}
double twin_hash_unit(uint64_t seed, uint32_t stream, uint64_t index) { return hash_unit(seed, stream, index); }

void* twin2d_create(int prec, int Nx, int Nz, const int npml[4], double dx, double dz, double dt, double mu0,
                    double eps0, int iJ, int kJ, int n_wf) {
    if (prec != 32 && prec != 64) return nullptr;
    if (Nx < 2 || Nz < 2 || npml[0] + npml[1] >= Nx || npml[2] + npml[3] >= Nz) return nullptr;
    auto* p = new Any2D;
    p->prec = prec;
    if (prec == 32) p->f.init(Nx, Nz, npml, dx, dz, dt, mu0, eps0, iJ, kJ, n_wf);
    else p->d.init(Nx, Nz, npml, dx, dz, dt, mu0, eps0, iJ, kJ, n_wf);
    return p;
}
void twin2d_destroy(void* h) { delete (Any2D*)h; }
size_t twin2d_size(void* h, int id) {
    auto* p = (Any2D*)h;
    if (p->prec == 32) { auto* v = find2(p->f, id); return v ? v->size() : 0; }
    auto* v = find2(p->d, id); return v ? v->size() : 0;
}
int twin2d_set(void* h, int id, const double* src, size_t n) {
    auto* p = (Any2D*)h;
    return p->prec == 32 ? put(find2(p->f, id), src, n) : put(find2(p->d, id), src, n);
}
int twin2d_get(void* h, int id, double* dst, size_t n) {
    auto* p = (Any2D*)h;
    return p->prec == 32 ? take(find2(p->f, id), dst, n) : take(find2(p->d, id), dst, n);
}
void twin2d_observe_accumulate(void* h, double weight, int reset) { auto* p = (Any2D*)h; if (p->prec == 32) p->f.observe_accumulate(weight, reset != 0); else p->d.observe_accumulate(weight, reset != 0); }
void twin2d_observe_rms(void* h, double* out) { auto* p = (Any2D*)h; if (p->prec == 32) p->f.observe_rms(out); else p->d.observe_rms(out); }
void twin2d_step(void* h, int n0, int count) {
    auto* p = (Any2D*)h;
    for (int n = n0; n < n0 + count; ++n) { if (p->prec == 32) p->f.step(n); else p->d.step(n); }
}

}  // extern "C"
