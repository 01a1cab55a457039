// setup_host.cpp -- fp64 host generators declared in include/fdtd_setup.h.
//
// Every expression keeps the operand order of the reference statement it restates, and libm is
// called exactly where the reference calls it (std::pow with an integer exponent included), so
// that on the same libm the tables are bit-identical to the reference's.
#include "../../include/fdtd_setup.h"

#include <cmath>
#include <vector>

namespace {
const double kC0 = 299792458.0;
const double kMu0 = 4.0 * M_PI * 1e-7;
const double kEps0 = 1.0 / (kC0 * kC0 * kMu0);
}  // namespace

extern "C" {

void fdtd_constants(double* c0, double* mu0, double* eps0, double* Z0) {
    if (c0) *c0 = kC0;
    if (mu0) *mu0 = kMu0;
    if (eps0) *eps0 = kEps0;
    if (Z0) *Z0 = std::sqrt(kMu0 / kEps0);
}

int fdtd_cpml_tables(int n, int side, double d_step, double d_den, double dt, double* out) {
    if (n < 0 || (n > 0 && !out) || !(d_step > 0) || !(d_den > 0) || !(dt > 0)) return 1;
    const int grading = 3;
    const int sigma_factor = 1;
    const double alpha_max = 0.05, kappa_max = 7.0, eps_r = 1.0, mu_r = 1.0;
    const double sigma_max = sigma_factor * (grading + 1) / (150.0 * M_PI * std::sqrt(eps_r * mu_r) * d_step);
    const double imp2 = kMu0 / kEps0;
    double* b_e = out;          double* a_e = out + n;     double* C_e = out + 2 * (size_t)n;
    double* b_m = out + 3 * (size_t)n; double* a_m = out + 4 * (size_t)n; double* C_m = out + 5 * (size_t)n;
    for (int l = 0; l < n; ++l) {
        // normalised depth of the electric and magnetic sample points of layer l
        const double re = side == 0 ? ((n - 0.75) - l) / n : (0.25 + l) / n;
        const double rm = side == 0 ? ((n - 0.25) - l) / n : (0.75 + l) / n;
        const double se = sigma_max * std::pow(re, grading);
        const double sm = imp2 * sigma_max * std::pow(rm, grading);
        const double ke = 1.0 + (kappa_max - 1.0) * std::pow(re, grading);
        const double km = 1.0 + (kappa_max - 1.0) * std::pow(rm, grading);
        const double ae = alpha_max * (1.0 - re);
        const double am = imp2 * alpha_max * (1.0 - rm);
        b_e[l] = std::exp(-(se / ke + ae) * dt / kEps0);
        b_m[l] = std::exp(-(sm / km + am) * dt / kMu0);
        const double de = d_den * ke * (se + ae * ke);
        const double dm = d_den * km * (sm + am * km);
        a_e[l] = std::abs(de) < 1e-30 ? 0.0 : se * (b_e[l] - 1.0) / de;
        a_m[l] = std::abs(dm) < 1e-30 ? 0.0 : sm * (b_m[l] - 1.0) / dm;
        C_e[l] = (dt / d_step) / ke;
        C_m[l] = (dt / d_step) / km;
    }
    return 0;
}

int fdtd3d_default_source(int Nx, int Ny, int n_wf, double dx, double dy, double dz, double dt, double* wf, double* mag) {
    if (Nx < 1 || Ny < 1 || n_wf < 0 || !wf || !mag) return 1;
    const double Z0 = std::sqrt(kMu0 / kEps0);
    const double wavelength = 1053e-9, E0 = 1.0;
    const double f0 = kC0 / wavelength, omega0 = 2.0 * M_PI * f0, T0 = 1.0 / f0;
    for (int n = 0; n < n_wf; ++n) {
        const double t = (n + 0.5) * dt;
        wf[n] = (2.0 * E0 / Z0) * std::sin(omega0 * t) / dz;
    }
    const double t_ramp = 5.0 * T0;
    const int n_ramp = static_cast<int>(std::round(t_ramp / dt));
    for (int n = 0; n <= n_ramp && n < n_wf; ++n) {
        const double s = static_cast<double>(n) / n_ramp;
        wf[n] *= 10.0 * std::pow(s, 3) - 15.0 * std::pow(s, 4) + 6.0 * std::pow(s, 5);
    }
    const double waist = 1.5 * wavelength;
    const int ic = Nx / 2, jc = Ny / 2;
    for (int i = 0; i < Nx; ++i) {
        const double x = (i - ic) * dx;
        for (int j = 0; j <= Ny; ++j) {
            const double y = (j - jc) * dy;
            mag[(size_t)i * (Ny + 1) + j] = E0 * std::exp(-(x * x + y * y) / (waist * waist));
        }
    }
    return 0;
}

int fdtd3d_default_materials(int Nx, int Ny, int Nz, double dx, double dy, double dz, double* eps_x, double* eps_y, double* eps_z) {
    if (Nx < 1 || Ny < 1 || Nz < 1 || !eps_x || !eps_y || !eps_z) return 1;
    const double n_glass = 1.47, n_air = 1.0;
    const double eps_glass = (n_glass * n_glass) * kEps0, eps_air = (n_air * n_air) * kEps0;
    const double radius = 2.0 * 1053e-9;
    const int N[3] = {Nx, Ny, Nz};
    const double h[3] = {dx, dy, dz};
    double* out[3] = {eps_x, eps_y, eps_z};
    for (int c = 0; c < 3; ++c) {
        int ext[3];
        for (int a = 0; a < 3; ++a) ext[a] = N[a] + (a != c);
        // component c sits half a cell along its own axis; the other two axes are sampled at the
        // node and half a cell either side of it
        const int u = (c + 1) % 3 < (c + 2) % 3 ? (c + 1) % 3 : (c + 2) % 3;
        const int v = 3 - c - u;
        for (int i = 0; i < ext[0]; ++i) for (int j = 0; j < ext[1]; ++j) for (int k = 0; k < ext[2]; ++k) {
            const int p[3] = {i, j, k};
            double r0[3];
            for (int a = 0; a < 3; ++a) r0[a] = a == c ? ((p[a] + 0.5) - N[a] / 2) * h[a] : (p[a] - N[a] / 2) * h[a];
            int hits = 0;
            for (int su = -1; su <= 1; ++su) for (int sv = -1; sv <= 1; ++sv) {
                double r[3] = {r0[0], r0[1], r0[2]};
                r[u] = r0[u] + su * (h[u] * 0.5);
                r[v] = r0[v] + sv * (h[v] * 0.5);
                if (std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]) <= radius) ++hits;
            }
            const double fill = static_cast<double>(hits) / 9.0;
            out[c][((size_t)i * ext[1] + j) * ext[2] + k] = eps_glass + (eps_air - eps_glass) * fill;
        }
    }
    return 0;
}

int fdtd2d_default_source(int n_wf, double dx, double dz, double dt, double* Jy) {
    if (n_wf < 0 || !Jy) return 1;
    const double f0 = 1.5e14, omega0 = 2.0 * M_PI * f0, T0 = 1.0 / f0, J0 = 1.0;
    for (int n = 0; n < n_wf; ++n) {
        const double t = (n + 1) * dt;
        Jy[n] = J0 * std::sin(omega0 * t) / (dx * dz) / (omega0 * kMu0 / 4.0);
    }
    const double t_ramp = 2.0 * T0;
    const int n_ramp = static_cast<int>(std::round(t_ramp / dt));
    for (int n = 0; n <= n_ramp && n < n_wf; ++n) {
        const double s = static_cast<double>(n) / n_ramp;
        Jy[n] *= s - std::sin(2.0 * M_PI * s) / (2.0 * M_PI);
    }
    return 0;
}

int fdtd2d_default_lorentz(int Nx, int Nz, double dt, double* C1, double* C2, double* C3) {
    if (Nx < 1 || Nz < 1 || !C1 || !C2 || !C3) return 1;
    for (size_t q = 0; q < (size_t)Nx * Nz; ++q) C1[q] = C2[q] = C3[q] = 0.0;
    const double f0 = 1.5e14, omega0 = 2.0 * M_PI * f0;
    const double w_plasma = omega0, w_res = w_plasma / std::sqrt(2.0), damping = 0.0;
    const double wp = w_plasma * dt, w0 = w_res * dt, dl = damping * dt;
    const int half = 40 / 2, mid = Nx / 2;
    for (int ix = mid - half; ix <= mid + half; ++ix) {
        if (ix < 0 || ix >= Nx) continue;          // the reference indexes unguarded; slabs wider than the grid are clipped here
        for (int kz = 0; kz <= Nz - 1; ++kz) {
            const size_t q = (size_t)ix * Nz + kz;
            const double den = (w0 * w0 + 4.0 * dl + 4.0);
            C1[q] = (wp * wp) / den;
            C2[q] = 2.0 * (w0 * w0 - 4.0) / den;
            C3[q] = (w0 * w0 - 4.0 * dl + 4.0) / den;
        }
    }
    return 0;
}

}  // extern "C"
