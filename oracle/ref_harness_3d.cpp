// ref_harness_3d.cpp -- drives the UNMODIFIED reference 3-D solver for parity pinning.
//
// TEST INFRASTRUCTURE ONLY, and only usable where the reference checkout is mounted
// (the build recipe in oracle/Makefile compiles /root/reference/fdtd3d/src/FDTD3D.cpp and
// OutputManager.cpp where they lie and links them with this file into
// oracle/_ref/libref3d.so; no reference source is copied into this repository).
//
// The reference keeps every array and every step function private
// (fdtd3d/include/FDTD3D.h:17-198), so the header is included with `private` opened up.
// Arrays are exchanged with the caller in the canonical layouts of include/fdtd_engine.h;
// the CPML auxiliary arrays are re-indexed through the reference's own id_Phi_* functions
// (FDTD3D.h:136-182) with the per-array shifts its sweeps apply (FDTD3D.cpp:546-1002).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <vector>
#include <unistd.h>

#define private public
#include "FDTD3D.h"
#undef private

#include "../include/fdtd_engine.h"

namespace {

struct Ref3 {
    FDTD3D* s;
};

std::vector<double>* field(FDTD3D* s, int id) {
    switch (id) {
        case FDTD3D_DX: return &s->Dx; case FDTD3D_DY: return &s->Dy; case FDTD3D_DZ: return &s->Dz;
        case FDTD3D_EX: return &s->Ex; case FDTD3D_EY: return &s->Ey; case FDTD3D_EZ: return &s->Ez;
        case FDTD3D_BX: return &s->Bx; case FDTD3D_BY: return &s->By; case FDTD3D_BZ: return &s->Bz;
        case FDTD3D_HX: return &s->Hx; case FDTD3D_HY: return &s->Hy; case FDTD3D_HZ: return &s->Hz;
        case FDTD3D_EPS_X: return &s->m_eps_x; case FDTD3D_EPS_Y: return &s->m_eps_y; case FDTD3D_EPS_Z: return &s->m_eps_z;
        case FDTD3D_JX_MAG: return &s->m_Jx_i_mag; case FDTD3D_JX_WF: return &s->m_wf_Jx_i;
    }
    return nullptr;
}

int layers(FDTD3D* s, int axis, int side) {
    const int n[3][2] = {{s->n_pml_xn, s->n_pml_xp}, {s->n_pml_yn, s->n_pml_yp}, {s->n_pml_zn, s->n_pml_zp}};
    return n[axis][side];
}

// kind: 0 b_e 1 a_e 2 C_e 3 b_m 4 a_m 5 C_m.  The reference stores C twice per face (one copy
// per affected component, FDTD3D.h:60-78); `second` selects the other copy for the self check.
std::vector<double>* table(FDTD3D* s, int axis, int side, int kind, bool second = false) {
    if (axis == 0) {
        if (side == 0) { std::vector<double>* t[] = {&s->b_e_xn, &s->a_e_xn, second ? &s->C_Dz_dxn_v : &s->C_Dy_dxn_v, &s->b_m_xn, &s->a_m_xn, second ? &s->C_Bz_dxn_v : &s->C_By_dxn_v}; return t[kind]; }
        std::vector<double>* t[] = {&s->b_e_xp, &s->a_e_xp, second ? &s->C_Dz_dxp_v : &s->C_Dy_dxp_v, &s->b_m_xp, &s->a_m_xp, second ? &s->C_Bz_dxp_v : &s->C_By_dxp_v}; return t[kind];
    }
    if (axis == 1) {
        if (side == 0) { std::vector<double>* t[] = {&s->b_e_yn, &s->a_e_yn, second ? &s->C_Dz_dyn_v : &s->C_Dx_dyn_v, &s->b_m_yn, &s->a_m_yn, second ? &s->C_Bz_dyn_v : &s->C_Bx_dyn_v}; return t[kind]; }
        std::vector<double>* t[] = {&s->b_e_yp, &s->a_e_yp, second ? &s->C_Dz_dyp_v : &s->C_Dx_dyp_v, &s->b_m_yp, &s->a_m_yp, second ? &s->C_Bz_dyp_v : &s->C_Bx_dyp_v}; return t[kind];
    }
    if (side == 0) { std::vector<double>* t[] = {&s->b_e_zn, &s->a_e_zn, second ? &s->C_Dy_dzn_v : &s->C_Dx_dzn_v, &s->b_m_zn, &s->a_m_zn, second ? &s->C_By_dzn_v : &s->C_Bx_dzn_v}; return t[kind]; }
    std::vector<double>* t[] = {&s->b_e_zp, &s->a_e_zp, second ? &s->C_Dy_dzp_v : &s->C_Dx_dzp_v, &s->b_m_zp, &s->a_m_zp, second ? &s->C_By_dzp_v : &s->C_Bx_dzp_v}; return t[kind];
}

struct PhiRef { std::vector<double>* v; long idx; };
// Reference storage slot of the auxiliary value belonging to cell (i,j,k) of component c,
// derivative axis a, face `side`, where the coordinate along `a` is the layer number.
// idx < 0: the reference never touches such a cell.
PhiRef phi_slot(FDTD3D* s, bool mag, int c, int a, int side, int i, int j, int k) {
    const int Nx = s->m_Nx, Ny = s->m_Ny, Nz = s->m_Nz;
    PhiRef r{nullptr, -1};
    if (!mag) {
        if (c == 0 && a == 1) { r.v = side ? &s->Phi_ex_yp : &s->Phi_ex_yn; if (i < Nx && k >= 1 && k < Nz) r.idx = side ? s->id_Phi_ex_yp(i, k - 1, j) : s->id_Phi_ex_yn(i, k - 1, j); }
        if (c == 0 && a == 2) { r.v = side ? &s->Phi_ex_zp : &s->Phi_ex_zn; if (i < Nx && j < Ny) r.idx = side ? s->id_Phi_ex_zp(i, j, k) : s->id_Phi_ex_zn(i, j, k); }
        if (c == 1 && a == 2) { r.v = side ? &s->Phi_ey_zp : &s->Phi_ey_zn; if (i >= 1 && i < Nx && j < Ny) r.idx = side ? s->id_Phi_ey_zp(i - 1, j, k) : s->id_Phi_ey_zn(i - 1, j, k); }
        if (c == 1 && a == 0) { r.v = side ? &s->Phi_ey_xp : &s->Phi_ey_xn; if (j < Ny && k >= 1 && k < Nz) r.idx = side ? s->id_Phi_ey_xp(i, j, k - 1) : s->id_Phi_ey_xn(i, j, k - 1); }
        if (c == 2 && a == 0) { r.v = side ? &s->Phi_ez_xp : &s->Phi_ez_xn; if (j >= 1 && j < Ny && k < Nz) r.idx = side ? s->id_Phi_ez_xp(i, j - 1, k) : s->id_Phi_ez_xn(i, j - 1, k); }
        if (c == 2 && a == 1) { r.v = side ? &s->Phi_ez_yp : &s->Phi_ez_yn; if (i >= 1 && i < Nx && k < Nz) r.idx = side ? s->id_Phi_ez_yp(i - 1, j, k) : s->id_Phi_ez_yn(i - 1, j, k); }
    } else {
        if (c == 0 && a == 1) { r.v = side ? &s->Phi_mx_yp : &s->Phi_mx_yn; if (i >= 1 && i < Nx && k < Nz) r.idx = side ? s->id_Phi_mx_yp(i - 1, j, k) : s->id_Phi_mx_yn(i - 1, j, k); }
        if (c == 0 && a == 2) { r.v = side ? &s->Phi_mx_zp : &s->Phi_mx_zn; if (i >= 1 && i < Nx && j < Ny) r.idx = side ? s->id_Phi_mx_zp(i - 1, j, k) : s->id_Phi_mx_zn(i - 1, j, k); }
        if (c == 1 && a == 0) { r.v = side ? &s->Phi_my_xp : &s->Phi_my_xn; if (j >= 1 && j < Ny && k < Nz) r.idx = side ? s->id_Phi_my_xp(i, j - 1, k) : s->id_Phi_my_xn(i, j - 1, k); }
        if (c == 1 && a == 2) { r.v = side ? &s->Phi_my_zp : &s->Phi_my_zn; if (i < Nx && j >= 1 && j < Ny) r.idx = side ? s->id_Phi_my_zp(i, j - 1, k) : s->id_Phi_my_zn(i, j - 1, k); }
        if (c == 2 && a == 0) { r.v = side ? &s->Phi_mz_xp : &s->Phi_mz_xn; if (j < Ny && k >= 1 && k < Nz) r.idx = side ? s->id_Phi_mz_xp(i, j, k - 1) : s->id_Phi_mz_xn(i, j, k - 1); }
        if (c == 2 && a == 1) { r.v = side ? &s->Phi_mz_yp : &s->Phi_mz_yn; if (i < Nx && k >= 1 && k < Nz) r.idx = side ? s->id_Phi_mz_yp(i, j, k - 1) : s->id_Phi_mz_yn(i, j, k - 1); }
    }
    return r;
}

void phi_box(FDTD3D* s, bool mag, int c, int a, int side, int n[3]) {
    const int N[3] = {s->m_Nx, s->m_Ny, s->m_Nz};
    for (int q = 0; q < 3; ++q) n[q] = mag ? N[q] + (q == c) : N[q] + (q != c);
    n[a] = layers(s, a, side);
}

bool decode_phi(int id, bool& mag, int& c, int& a, int& side) {
    int r;
    if (id >= FDTD3D_PHI_E && id < FDTD3D_PHI_E + 18) { mag = false; r = id - FDTD3D_PHI_E; }
    else if (id >= FDTD3D_PHI_M && id < FDTD3D_PHI_M + 18) { mag = true; r = id - FDTD3D_PHI_M; }
    else return false;
    side = r % 2; a = (r / 2) % 3; c = r / 6;
    return a != c;
}

}  // namespace

extern "C" {

// Constructs the reference solver with its hard-coded 200^3 problem.  Its OutputManager wipes
// and recreates ./output_3d in the current directory (fdtd3d/src/OutputManager.cpp:13-17), so the
// process first changes into `scratch_dir`.
void* ref3d_create(const char* scratch_dir) {
    if (scratch_dir && chdir(scratch_dir) != 0) return nullptr;
    auto* r = new Ref3;
    r->s = new FDTD3D();
    return r;
}
void ref3d_destroy(void* h) { auto* r = (Ref3*)h; delete r->s; delete r; }

// Re-parameterise without touching the reference: overwrite the size members, re-assign the 15
// arrays setupGrid() sized (FDTD3D.cpp:42-63), then let the reference's own setupSource /
// setupMaterials / setupPML rebuild everything else from the members.
int ref3d_reshape(void* h, int Nx, int Ny, int Nz, const int npml[6], int steps) {
    FDTD3D* s = ((Ref3*)h)->s;
    if (Nz < 21) return 1;                         // setupSource() pins the source plane to k = 20
    s->m_Nx = Nx; s->m_Ny = Ny; s->m_Nz = Nz; s->m_steps = steps;
    s->n_pml_xn = npml[0]; s->n_pml_xp = npml[1]; s->n_pml_yn = npml[2];
    s->n_pml_yp = npml[3]; s->n_pml_zn = npml[4]; s->n_pml_zp = npml[5];
    s->Dx.assign((size_t)Nx * (Ny + 1) * (Nz + 1), 0.0); s->Ex = s->Dx; s->m_eps_x = s->Dx;
    s->Dy.assign((size_t)(Nx + 1) * Ny * (Nz + 1), 0.0); s->Ey = s->Dy; s->m_eps_y = s->Dy;
    s->Dz.assign((size_t)(Nx + 1) * (Ny + 1) * Nz, 0.0); s->Ez = s->Dz; s->m_eps_z = s->Dz;
    s->Bx.assign((size_t)(Nx + 1) * Ny * Nz, 0.0); s->Hx = s->Bx;
    s->By.assign((size_t)Nx * (Ny + 1) * Nz, 0.0); s->Hy = s->By;
    s->Bz.assign((size_t)Nx * Ny * (Nz + 1), 0.0); s->Hz = s->Bz;
    s->setupSource();
    s->setupMaterials();
    s->setupPML();
    return 0;
}

void ref3d_scalars(void* h, double out[8], int iout[8]) {
    FDTD3D* s = ((Ref3*)h)->s;
    out[0] = s->m_dx; out[1] = s->m_dy; out[2] = s->m_dz; out[3] = s->m_dt;
    out[4] = s->m_mu0; out[5] = s->m_eps0; out[6] = s->m_c0; out[7] = s->m_Z0;
    iout[0] = s->m_Nx; iout[1] = s->m_Ny; iout[2] = s->m_Nz; iout[3] = s->m_steps; iout[4] = s->m_k_Jx;
    iout[5] = iout[6] = iout[7] = 0;
}
void ref3d_set_kjx(void* h, int k) { ((Ref3*)h)->s->m_k_Jx = k; }

size_t ref3d_size(void* h, int id) {
    FDTD3D* s = ((Ref3*)h)->s;
    if (auto* v = field(s, id)) return v->size();
    bool mag; int c, a, side;
    if (decode_phi(id, mag, c, a, side)) { int n[3]; phi_box(s, mag, c, a, side, n); return (size_t)n[0] * n[1] * n[2]; }
    if (id >= FDTD3D_TAB && id < FDTD3D_TAB + 36) { int r = id - FDTD3D_TAB; return layers(s, r / 12, (r / 6) % 2); }
    return 0;
}

int ref3d_get(void* h, int id, double* dst, size_t n) {
    FDTD3D* s = ((Ref3*)h)->s;
    if (ref3d_size(h, id) != n) return 1;
    if (auto* v = field(s, id)) { std::memcpy(dst, v->data(), n * sizeof(double)); return 0; }
    bool mag; int c, a, side;
    if (decode_phi(id, mag, c, a, side)) {
        int b[3]; phi_box(s, mag, c, a, side, b);
        size_t o = 0;
        for (int i = 0; i < b[0]; ++i) for (int j = 0; j < b[1]; ++j) for (int k = 0; k < b[2]; ++k, ++o) {
            PhiRef r = phi_slot(s, mag, c, a, side, i, j, k);
            dst[o] = (r.v && r.idx >= 0 && (size_t)r.idx < r.v->size()) ? (*r.v)[r.idx] : 0.0;
        }
        return 0;
    }
    if (id >= FDTD3D_TAB && id < FDTD3D_TAB + 36) {
        int r = id - FDTD3D_TAB;
        auto* v = table(s, r / 12, (r / 6) % 2, r % 6);
        for (size_t q = 0; q < n; ++q) dst[q] = (*v)[q];
        return 0;
    }
    return 1;
}

// One entry of a field array (flat index in the reference's own order), for per-step probe series.
double ref3d_peek(void* h, int id, size_t flat) {
    auto* v = field(((Ref3*)h)->s, id);
    return (v && flat < v->size()) ? (*v)[flat] : 0.0;
}

int ref3d_set(void* h, int id, const double* src, size_t n) {
    FDTD3D* s = ((Ref3*)h)->s;
    if (ref3d_size(h, id) != n) return 1;
    if (auto* v = field(s, id)) { std::memcpy(v->data(), src, n * sizeof(double)); return 0; }
    bool mag; int c, a, side;
    if (decode_phi(id, mag, c, a, side)) {
        int b[3]; phi_box(s, mag, c, a, side, b);
        size_t o = 0;
        for (int i = 0; i < b[0]; ++i) for (int j = 0; j < b[1]; ++j) for (int k = 0; k < b[2]; ++k, ++o) {
            PhiRef r = phi_slot(s, mag, c, a, side, i, j, k);
            if (r.v && r.idx >= 0 && (size_t)r.idx < r.v->size()) (*r.v)[r.idx] = src[o];
        }
        return 0;
    }
    if (id >= FDTD3D_TAB && id < FDTD3D_TAB + 36) {
        int r = id - FDTD3D_TAB;
        for (int second = 0; second < 2; ++second) {
            auto* v = table(s, r / 12, (r / 6) % 2, r % 6, second != 0);
            for (size_t q = 0; q < n; ++q) (*v)[q] = src[q];
        }
        return 0;
    }
    return 1;
}

// 0 if the duplicated C tables of every face agree (so exporting one of them loses nothing).
int ref3d_tables_consistent(void* h) {
    FDTD3D* s = ((Ref3*)h)->s;
    for (int a = 0; a < 3; ++a) for (int sd = 0; sd < 2; ++sd) for (int kind : {2, 5}) {
        auto* u = table(s, a, sd, kind, false); auto* v = table(s, a, sd, kind, true);
        int n = layers(s, a, sd);
        for (int q = 0; q < n; ++q) if ((*u)[q] != (*v)[q]) return 1;
    }
    return 0;
}

void ref3d_step(void* h, int n0, int count) {
    FDTD3D* s = ((Ref3*)h)->s;
    for (int n = n0; n < n0 + count; ++n) { s->updateE(n); s->updateH(); }     // FDTD3D.cpp:1085-1086
}
// The reference's whole time loop with its RMS window and file output (FDTD3D.cpp:1062-1111);
// files land in <scratch_dir>/output_3d.
void ref3d_run(void* h) { ((Ref3*)h)->s->run(); }
void ref3d_half_e(void* h, int n) { ((Ref3*)h)->s->updateE(n); }
void ref3d_half_h(void* h) { ((Ref3*)h)->s->updateH(); }

}  // extern "C"
