/*
 * fdtd_setup.h -- host-side generators of the arrays the step kernels consume.
 *
 * These restate, in fp64, the once-per-run setup code of the reference solvers so that a host
 * driver can build the reference's default problems (or scaled versions of them) and hand the
 * results to the engine through fdtd_engine.h.  Plain C ABI, no device code.
 */
#ifndef FDTD_SETUP_H
#define FDTD_SETUP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* c0, mu0 = 4 pi 1e-7, eps0 = 1/(c0^2 mu0), Z0 = sqrt(mu0/eps0)   (FDTD3D.h:19-22, FDTD2D.h:18-20) */
void fdtd_constants(double* c0, double* mu0, double* eps0, double* Z0);

/* One CPML face: out[kind*n + l], kind 0 b_e 1 a_e 2 C_e 3 b_m 4 a_m 5 C_m, l = 0..n-1.
 * side 0 = low face, 1 = high face.  d_step enters sigma_max and C, d_den the denominators of
 * a_e/a_m: equal in 3-D (FDTD3D.cpp:226-417); the 2-D x faces pass d_step = dx, d_den = dz
 * (FDTD2D.cpp:161-162, :192-193).  Polynomial grading m = 3, kappa_max = 7, alpha_max = 0.05. */
int fdtd_cpml_tables(int n, int side, double d_step, double d_den, double dt, double* out);

/* FDTD3D::setupSource (FDTD3D.cpp:69-104): ramped sine wf[n_wf] and Gaussian sheet mag[Nx][Ny+1]. */
int fdtd3d_default_source(int Nx, int Ny, int n_wf, double dx, double dy, double dz, double dt,
                          double* wf, double* mag);
/* FDTD3D::setupMaterials (FDTD3D.cpp:106-200): fused silica with a centred air sphere of radius
 * two wavelengths, 3x3 sub-sampled; arrays in canonical order. */
int fdtd3d_default_materials(int Nx, int Ny, int Nz, double dx, double dy, double dz,
                             double* eps_x, double* eps_y, double* eps_z);
/* FDTD2D::setupSource (FDTD2D.cpp:58-82). */
int fdtd2d_default_source(int n_wf, double dx, double dz, double dt, double* Jy);
/* FDTD2D::setupMaterialsLorentz (FDTD2D.cpp:84-112): slab ix in [Nx/2-20, Nx/2+20]; C arrays [Nx][Nz]. */
int fdtd2d_default_lorentz(int Nx, int Nz, double dt, double* C1, double* C2, double* C3);

#ifdef __cplusplus
}
#endif
#endif
