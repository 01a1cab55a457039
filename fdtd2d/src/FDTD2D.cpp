// FDTD2D.cpp -- host driver: setup through fdtd_setup.h, stepping and the RMS window through
// fdtd_engine.h, output cadence of the reference's run() (fdtd2d/src/FDTD2D.cpp:501-535).
#include "FDTD2D.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <stdexcept>

#include "FieldWriter2D.h"
#include "fdtd_engine.h"
#include "fdtd_setup.h"

namespace {
void must(int rc, const char* what) {
    if (rc != 0) throw std::runtime_error(std::string(what) + ": " + fdtd_last_error());
}
bool env_int(const char* name, int& v) {
    const char* e = std::getenv(name);
    if (!e || !*e) return false;
    v = std::atoi(e);
    return true;
}
std::string numbered(const char* stem, int step) {
    std::ostringstream s;
    s << stem << std::setw(5) << std::setfill('0') << step << ".csv";
    return s.str();
}
}  // namespace

Sim2DConfig Sim2DConfig::reference_default() {
    double c0;
    fdtd_constants(&c0, nullptr, nullptr, nullptr);
    Sim2DConfig c{};
    const double f0 = 1.5e14, wavelength = c0 / f0;
    const int cells_per_wavelength = 40;
    c.dx = wavelength / cells_per_wavelength;
    c.dz = c.dx;
    c.Nx = static_cast<int>(std::round(5.0 * wavelength / c.dx));
    c.Nz = static_cast<int>(std::round(12.0 * wavelength / c.dz));
    const double cfl = 2.0;
    c.dt = c.dx / (c0 * cfl);
    c.steps = 1580;
    for (int& n : c.n_pml) n = 20;
    c.i_Jy = c.Nx / 2 - cells_per_wavelength;
    c.k_Jy = c.Nz / 2;
    c.slab = 1;
    c.precision = 64;
    c.fused = true;
    c.step_by_step = false;
    c.n_T0 = static_cast<int>(std::round(cfl * cells_per_wavelength));
    c.frame_every = 20;
    c.write_files = true;
    c.out_dir = "output_2d";
    return c;
}

void Sim2DConfig::apply_environment() {
    const bool resized = env_int("FDTD2D_NX", Nx) | env_int("FDTD2D_NZ", Nz);
    if (resized) { i_Jy = Nx / 2 - 40; k_Jy = Nz / 2; }      // the reference derives the source cell from the grid
    env_int("FDTD2D_STEPS", steps); env_int("FDTD2D_IJY", i_Jy); env_int("FDTD2D_KJY", k_Jy);
    env_int("FDTD2D_SLAB", slab); env_int("FDTD2D_PRECISION", precision); env_int("FDTD2D_FRAME_EVERY", frame_every);
    int v;
    if (env_int("FDTD2D_NPML", v)) for (int& n : n_pml) n = v;
    if (env_int("FDTD2D_WRITE", v)) write_files = v != 0;
    if (env_int("FDTD2D_FUSED", v)) fused = v != 0;
    if (env_int("FDTD2D_STEP_BY_STEP", v)) step_by_step = v != 0;
    if (const char* e = std::getenv("FDTD2D_OUT")) if (*e) out_dir = e;
}

FDTD2D::FDTD2D() : m_cfg(Sim2DConfig::reference_default()) {
    m_cfg.apply_environment();
    build();
}
FDTD2D::FDTD2D(const Sim2DConfig& cfg) : m_cfg(cfg) { build(); }
FDTD2D::~FDTD2D() { fdtd2d_destroy(m_engine); }

void FDTD2D::build() {
    const Sim2DConfig& c = m_cfg;
    std::cout << "[grid] Nx=" << c.Nx << ", Nz=" << c.Nz << ", dx=" << c.dx << ", dz=" << c.dz << ", dt=" << c.dt
              << ", steps=" << c.steps << "\n";
    double mu0, eps0;
    fdtd_constants(nullptr, &mu0, &eps0, nullptr);
    fdtd2d_desc d{};
    d.struct_size = sizeof d;
    d.Nx = c.Nx; d.Nz = c.Nz;
    for (int q = 0; q < 4; ++q) d.n_pml[q] = c.n_pml[q];
    // a source cell outside the grid is silently ignored by the reference (FDTD2D.cpp:360): park it off-grid
    const bool source_on_grid = c.i_Jy >= 0 && c.i_Jy <= c.Nx && c.k_Jy >= 0 && c.k_Jy <= c.Nz;
    d.i_Jy = source_on_grid ? c.i_Jy : -1; d.k_Jy = source_on_grid ? c.k_Jy : -1;
    d.n_wf = c.steps; d.precision = c.precision; d.device = -1;
    d.mode = c.fused ? FDTD2D_MODE_FUSED : FDTD2D_MODE_FULL;
    d.dx = c.dx; d.dz = c.dz; d.dt = c.dt; d.mu0 = mu0; d.eps0 = eps0;
    must(fdtd2d_create(&d, &m_engine), "fdtd2d_create");

    std::vector<double> jy(c.steps);
    must(fdtd2d_default_source(c.steps, c.dx, c.dz, c.dt, jy.data()), "default source");
    must(fdtd2d_set_array(m_engine, FDTD2D_JY, jy.data(), jy.size()), "upload Jy");
    if (c.slab) {
        const size_t n = static_cast<size_t>(c.Nx) * c.Nz;
        std::vector<double> c1(n), c2(n), c3(n);
        must(fdtd2d_default_lorentz(c.Nx, c.Nz, c.dt, c1.data(), c2.data(), c3.data()), "Lorentz coefficients");
        must(fdtd2d_set_array(m_engine, FDTD2D_C1, c1.data(), n), "upload C1");
        must(fdtd2d_set_array(m_engine, FDTD2D_C2, c2.data(), n), "upload C2");
        must(fdtd2d_set_array(m_engine, FDTD2D_C3, c3.data(), n), "upload C3");
    }
    const double h[2] = {c.dx, c.dz};
    for (int axis = 0; axis < 2; ++axis) for (int side = 0; side < 2; ++side) {
        const int n = c.n_pml[axis * 2 + side];
        if (n == 0) continue;
        std::vector<double> t(static_cast<size_t>(6) * n);
        // the x faces divide their a coefficients by dz, as the reference does (FDTD2D.cpp:161-162)
        must(fdtd_cpml_tables(n, side, h[axis], c.dz, c.dt, t.data()), "CPML tables");
        for (int kind = 0; kind < 6; ++kind)
            must(fdtd2d_set_array(m_engine, FDTD2D_TAB + (axis * 2 + side) * 6 + kind, t.data() + static_cast<size_t>(kind) * n, n), "upload CPML table");
    }
    if (c.write_files) m_out = std::make_unique<FieldWriter2D>(c.out_dir);
    m_ey.assign(fdtd2d_array_size(m_engine, FDTD2D_EY), 0.0);
    m_center.assign(static_cast<size_t>(c.Nx) * c.Nz, 0.0);
}

void FDTD2D::dump_frame(int step) {
    const Sim2DConfig& c = m_cfg;
    must(fdtd2d_get_array(m_engine, FDTD2D_EY, m_ey.data(), m_ey.size()), "download Ey");
    for (int i = 0; i < c.Nx; ++i)
        for (int k = 0; k < c.Nz; ++k) m_center[static_cast<size_t>(i) * c.Nz + k] = m_ey[static_cast<size_t>(i) * (c.Nz + 1) + k];
    if (!m_out) return;
    m_out->write_csv(numbered("ey_t", step), m_center, c.Nx, c.Nz, c.dx, c.dz);
    if (step == 0) m_out->write_dims(c.Nx, c.Nz, c.dx, c.dz);
}

void FDTD2D::run() {
    const Sim2DConfig& c = m_cfg;
    const int half_period = c.n_T0 / 2;
    const double weight = 4.0 / c.n_T0;
    // The reference looks at the fields after every step (FDTD2D.cpp:510-531): the RMS window is advanced, a completed
    // window written out, every frame_every-th step dumped.  Here the steps between two of those moments go to the engine
    // in one call, which advances the window on the device by itself (two steps per launch on large grids).  A run of
    // steps ends at the next step whose field is dumped, and one step after the next step that completes a window at
    // the latest -- the engine keeps a completed window until the next one completes.
    const int every = c.frame_every > 0 ? c.frame_every : c.steps + 1;
    for (int n = 0; n < c.steps;) {
        const auto t0 = std::chrono::steady_clock::now();
        const int next_dump = (n + every - 1) / every * every;                                   // first step >= n that is dumped
        const int next_done = (n + half_period) / half_period * half_period - 1;                 // first step >= n that completes a window
        const int last = c.step_by_step ? n : std::min(std::min(next_dump, next_done + (half_period > 1 ? 1 : 0)), c.steps - 1);
        must(fdtd2d_step_observed(m_engine, n, last - n + 1, weight, half_period, nullptr), "step");
        must(fdtd2d_sync(m_engine, nullptr), "sync");
        m_device_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (next_done <= last) {
            must(fdtd2d_observe_rms_window(m_engine, m_center.data(), m_center.size()), "RMS download");
            if (m_out) m_out->write_csv(numbered("eymag_t", next_done + 1), m_center, c.Nx, c.Nz, c.dx, c.dz);
        }
        if (last % every == 0) {
            dump_frame(last);
            std::cout << "[n=" << last << "] frame dumped.\n";
        }
        n = last + 1;
        m_step_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    dump_frame(c.steps);
    std::cout << "Done. CSV in ./" << c.out_dir << "\n";
}
