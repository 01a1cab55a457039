// ref_harness_2d.cpp -- drives the UNMODIFIED reference 2-D solver for parity pinning.
//
// TEST INFRASTRUCTURE ONLY; see ref_harness_3d.cpp for the rules.  Built by oracle/Makefile from
// /root/reference/fdtd2d/src/{FDTD2D,OutputManager}.cpp into oracle/_ref/libref2d.so.
// State and step function are private in the reference (fdtd2d/include/FDTD2D.h:16-92).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <vector>
#include <unistd.h>

#define private public
#include "FDTD2D.h"
#undef private

#include "../include/fdtd_engine.h"

namespace {

std::vector<double>* field(FDTD2D* s, int id) {
    switch (id) {
        case FDTD2D_DY: return &s->Dy; case FDTD2D_EY: return &s->Ey; case FDTD2D_EY_N1: return &s->Ey_n1; case FDTD2D_EY_N2: return &s->Ey_n2;
        case FDTD2D_SEY: return &s->SEy; case FDTD2D_SEY_N1: return &s->SEy_n1; case FDTD2D_SEY_N2: return &s->SEy_n2;
        case FDTD2D_BX: return &s->Bx; case FDTD2D_HX: return &s->Hx; case FDTD2D_HX_N1: return &s->Hx_n1; case FDTD2D_HX_N2: return &s->Hx_n2;
        case FDTD2D_SHX: return &s->SHx; case FDTD2D_SHX_N1: return &s->SHx_n1; case FDTD2D_SHX_N2: return &s->SHx_n2;
        case FDTD2D_BZ: return &s->Bz; case FDTD2D_HZ: return &s->Hz; case FDTD2D_HZ_N1: return &s->Hz_n1; case FDTD2D_HZ_N2: return &s->Hz_n2;
        case FDTD2D_SHZ: return &s->SHz; case FDTD2D_SHZ_N1: return &s->SHz_n1; case FDTD2D_SHZ_N2: return &s->SHz_n2;
        case FDTD2D_C1: return &s->C1; case FDTD2D_C2: return &s->C2; case FDTD2D_C3: return &s->C3;
        case FDTD2D_JY: return &s->Jy;
    }
    return nullptr;
}
int layers(FDTD2D* s, int axis, int side) {
    const int n[2][2] = {{s->n_pml_xn, s->n_pml_xp}, {s->n_pml_zn, s->n_pml_zp}};
    return n[axis][side];
}
// axis 0 = x, 1 = z.  kind: 0 b_e 1 a_e 2 C_e 3 b_m 4 a_m 5 C_m  (FDTD2D.h:43-51)
std::vector<double>* table(FDTD2D* s, int axis, int side, int kind) {
    if (axis == 0) {
        if (side == 0) { std::vector<double>* t[] = {&s->b_e_xn, &s->a_e_xn, &s->C_Dy_dxn_v, &s->b_m_xn, &s->a_m_xn, &s->C_Bz_dxn_v}; return t[kind]; }
        std::vector<double>* t[] = {&s->b_e_xp, &s->a_e_xp, &s->C_Dy_dxp_v, &s->b_m_xp, &s->a_m_xp, &s->C_Bz_dxp_v}; return t[kind];
    }
    if (side == 0) { std::vector<double>* t[] = {&s->b_e_zn, &s->a_e_zn, &s->C_Dy_dzn_v, &s->b_m_zn, &s->a_m_zn, &s->C_Bx_dzn_v}; return t[kind]; }
    std::vector<double>* t[] = {&s->b_e_zp, &s->a_e_zp, &s->C_Dy_dzp_v, &s->b_m_zp, &s->a_m_zp, &s->C_Bx_dzp_v}; return t[kind];
}

struct PhiRef { std::vector<double>* v; long idx; int n0, n1; };
// which: 0 ey_z, 1 ey_x, 2 mx_z, 3 mz_x.  Canonical shapes [Nx+1][n] (z kinds, index (i,l))
// and [n][Nz+1] (x kinds, index (l,k)); reference slots via id_x_pml / id_z_pml with the
// shifts used in FDTD2D.cpp:290-295, :325-330.
PhiRef phi_slot(FDTD2D* s, int which, int side, int u, int v) {
    const int Nx = s->m_Nx, Nz = s->m_Nz;
    PhiRef r{nullptr, -1, 0, 0};
    if (which == 0 || which == 2) {
        const int n = layers(s, 1, side);
        r.v = which == 0 ? (side ? &s->Phi_ey_zp : &s->Phi_ey_zn) : (side ? &s->Phi_mx_zp : &s->Phi_mx_zn);
        r.n0 = Nx + 1; r.n1 = n;
        if (u >= 1 && u <= Nx - 1) r.idx = s->id_x_pml(u - 1, v, n);
    } else {
        const int n = layers(s, 0, side);
        r.v = which == 1 ? (side ? &s->Phi_ey_xp : &s->Phi_ey_xn) : (side ? &s->Phi_mz_xp : &s->Phi_mz_xn);
        r.n0 = n; r.n1 = Nz + 1;
        if (v >= 1 && v <= Nz - 1) r.idx = s->id_z_pml(u, v - 1, Nz - 1);
    }
    return r;
}

}  // namespace

extern "C" {

// The reference constructor builds its hard-coded 200x480 problem and its OutputManager wipes
// ./output_2d (fdtd2d/src/OutputManager.cpp:8-18): run inside `scratch_dir`.
void* ref2d_create(const char* scratch_dir) {
    if (scratch_dir && chdir(scratch_dir) != 0) return nullptr;
    return new FDTD2D();
}
void ref2d_destroy(void* h) { delete (FDTD2D*)h; }

int ref2d_reshape(void* h, int Nx, int Nz, const int npml[4], int steps) {
    FDTD2D* s = (FDTD2D*)h;
    if (Nx < 42) return 1;             // setupMaterialsLorentz() indexes Nx/2 +- 20 (FDTD2D.cpp:86-90)
    s->m_Nx = Nx; s->m_Nz = Nz; s->m_steps = steps;
    s->n_pml_xn = npml[0]; s->n_pml_xp = npml[1]; s->n_pml_zn = npml[2]; s->n_pml_zp = npml[3];
    s->Dy.assign((size_t)(Nx + 1) * (Nz + 1), 0.0);
    s->Ey = s->Dy; s->Ey_n1 = s->Dy; s->Ey_n2 = s->Dy; s->SEy = s->Dy; s->SEy_n1 = s->Dy; s->SEy_n2 = s->Dy;
    s->Bx.assign((size_t)(Nx + 1) * Nz, 0.0);
    s->Hx = s->Bx; s->Hx_n1 = s->Bx; s->Hx_n2 = s->Bx; s->SHx = s->Bx; s->SHx_n1 = s->Bx; s->SHx_n2 = s->Bx;
    s->Bz.assign((size_t)Nx * (Nz + 1), 0.0);
    s->Hz = s->Bz; s->Hz_n1 = s->Bz; s->Hz_n2 = s->Bz; s->SHz = s->Bz; s->SHz_n1 = s->Bz; s->SHz_n2 = s->Bz;
    s->C1.assign((size_t)Nx * Nz, 0.0); s->C2 = s->C1; s->C3 = s->C1;
    s->setupSource();
    s->setupMaterialsLorentz();
    s->setupPML();
    return 0;
}

void ref2d_scalars(void* h, double out[8], int iout[8]) {
    FDTD2D* s = (FDTD2D*)h;
    out[0] = s->m_dx; out[1] = s->m_dz; out[2] = s->m_dt; out[3] = s->m_mu0; out[4] = s->m_eps0; out[5] = s->m_c0;
    out[6] = out[7] = 0;
    iout[0] = s->m_Nx; iout[1] = s->m_Nz; iout[2] = s->m_steps; iout[3] = s->i_Jy; iout[4] = s->k_Jy;
    iout[5] = iout[6] = iout[7] = 0;
}
void ref2d_set_source_cell(void* h, int i, int k) { FDTD2D* s = (FDTD2D*)h; s->i_Jy = i; s->k_Jy = k; }

size_t ref2d_size(void* h, int id) {
    FDTD2D* s = (FDTD2D*)h;
    if (auto* v = field(s, id)) return v->size();
    if (id >= FDTD2D_PHI && id < FDTD2D_PHI + 8) { PhiRef r = phi_slot(s, (id - FDTD2D_PHI) / 2, (id - FDTD2D_PHI) % 2, 0, 0); return (size_t)r.n0 * r.n1; }
    if (id >= FDTD2D_TAB && id < FDTD2D_TAB + 24) { int r = id - FDTD2D_TAB; return layers(s, r / 12, (r / 6) % 2); }
    return 0;
}
// One entry of a field array (flat index in the reference's own order), for per-step probe series.
double ref2d_peek(void* h, int id, size_t flat) {
    auto* v = field((FDTD2D*)h, id);
    return (v && flat < v->size()) ? (*v)[flat] : 0.0;
}
int ref2d_get(void* h, int id, double* dst, size_t n) {
    FDTD2D* s = (FDTD2D*)h;
    if (ref2d_size(h, id) != n) return 1;
    if (auto* v = field(s, id)) { std::memcpy(dst, v->data(), n * sizeof(double)); return 0; }
    if (id >= FDTD2D_PHI && id < FDTD2D_PHI + 8) {
        const int which = (id - FDTD2D_PHI) / 2, side = (id - FDTD2D_PHI) % 2;
        PhiRef b = phi_slot(s, which, side, 0, 0);
        size_t o = 0;
        for (int u = 0; u < b.n0; ++u) for (int v = 0; v < b.n1; ++v, ++o) {
            PhiRef r = phi_slot(s, which, side, u, v);
            dst[o] = (r.idx >= 0 && (size_t)r.idx < r.v->size()) ? (*r.v)[r.idx] : 0.0;
        }
        return 0;
    }
    if (id >= FDTD2D_TAB && id < FDTD2D_TAB + 24) {
        int r = id - FDTD2D_TAB;
        auto* v = table(s, r / 12, (r / 6) % 2, r % 6);
        for (size_t q = 0; q < n; ++q) dst[q] = (*v)[q];
        return 0;
    }
    return 1;
}
int ref2d_set(void* h, int id, const double* src, size_t n) {
    FDTD2D* s = (FDTD2D*)h;
    if (ref2d_size(h, id) != n) return 1;
    if (auto* v = field(s, id)) { std::memcpy(v->data(), src, n * sizeof(double)); return 0; }
    if (id >= FDTD2D_PHI && id < FDTD2D_PHI + 8) {
        const int which = (id - FDTD2D_PHI) / 2, side = (id - FDTD2D_PHI) % 2;
        PhiRef b = phi_slot(s, which, side, 0, 0);
        size_t o = 0;
        for (int u = 0; u < b.n0; ++u) for (int v = 0; v < b.n1; ++v, ++o) {
            PhiRef r = phi_slot(s, which, side, u, v);
            if (r.idx >= 0 && (size_t)r.idx < r.v->size()) (*r.v)[r.idx] = src[o];
        }
        return 0;
    }
    if (id >= FDTD2D_TAB && id < FDTD2D_TAB + 24) {
        int r = id - FDTD2D_TAB;
        auto* v = table(s, r / 12, (r / 6) % 2, r % 6);
        for (size_t q = 0; q < n; ++q) (*v)[q] = src[q];
        return 0;
    }
    return 1;
}
// The reference's whole time loop with RMS window and CSV output (FDTD2D.cpp:501-535); files land
// in <scratch_dir>/output_2d.
void ref2d_run(void* h) { ((FDTD2D*)h)->run(); }
void ref2d_step(void* h, int n0, int count) {
    FDTD2D* s = (FDTD2D*)h;
    for (int n = n0; n < n0 + count; ++n) s->stepOnce(n);       // FDTD2D.cpp:512
}

}  // extern "C"
