// FDTD3D.cpp -- host driver: setup through fdtd_setup.h, stepping and observables through
// fdtd_engine.h, output cadence of the reference's run() (fdtd3d/src/FDTD3D.cpp:1062-1111).
#include "FDTD3D.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <stdexcept>

#include "FieldWriter3D.h"
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
}  // namespace

Sim3DConfig Sim3DConfig::reference_default() {
    double c0;
    fdtd_constants(&c0, nullptr, nullptr, nullptr);
    Sim3DConfig c{};
    const double wavelength = 1053e-9;
    const int cells_per_wavelength = 20;
    c.dx = c.dy = c.dz = wavelength / cells_per_wavelength;
    c.Nx = static_cast<int>(std::round(10.0 * wavelength / c.dx));
    c.Ny = static_cast<int>(std::round(10.0 * wavelength / c.dy));
    c.Nz = static_cast<int>(std::round(10.0 * wavelength / c.dz));
    const double cfl = 2.0;
    c.dt = c.dx / (c0 * cfl);
    c.steps = 1501;
    for (int& n : c.n_pml) n = 20;
    c.k_Jx = 20;
    c.precision = 64;
    c.lean = true;
    c.fused = true;
    c.slabs = 1;
    c.binary_vti = false;
    c.n_T0 = static_cast<int>(std::round(2 * cells_per_wavelength));
    c.frame_every = 20;
    c.write_files = true;
    c.step_by_step = false;
    c.out_dir = "output_3d";
    return c;
}

void Sim3DConfig::apply_environment() {
    env_int("FDTD3D_NX", Nx); env_int("FDTD3D_NY", Ny); env_int("FDTD3D_NZ", Nz);
    env_int("FDTD3D_STEPS", steps); env_int("FDTD3D_KJX", k_Jx); env_int("FDTD3D_PRECISION", precision);
    env_int("FDTD3D_FRAME_EVERY", frame_every);
    int v;
    if (env_int("FDTD3D_NPML", v)) for (int& n : n_pml) n = v;
    if (env_int("FDTD3D_LEAN", v)) lean = v != 0;
    if (env_int("FDTD3D_FUSED", v)) fused = v != 0;
    env_int("FDTD3D_SLABS", slabs);
    if (env_int("FDTD3D_WRITE", v)) write_files = v != 0;
    if (env_int("FDTD3D_STEP_BY_STEP", v)) step_by_step = v != 0;
    if (env_int("FDTD3D_BINARY_VTI", v)) binary_vti = v != 0;
    if (const char* e = std::getenv("FDTD3D_OUT")) if (*e) out_dir = e;
}

FDTD3D::FDTD3D() : m_cfg(Sim3DConfig::reference_default()) {
    m_cfg.apply_environment();
    build();
}
FDTD3D::FDTD3D(const Sim3DConfig& cfg) : m_cfg(cfg) { build(); }

FDTD3D::~FDTD3D() { fdtd3d_group_destroy(m_group); }

void FDTD3D::build() {
    const Sim3DConfig& c = m_cfg;
    std::cout << "[grid] Nx=" << c.Nx << ", Ny=" << c.Ny << ", Nz=" << c.Nz << ", dx=" << c.dx << ", dt=" << c.dt
              << ", steps=" << c.steps << "\n";
    double mu0;
    fdtd_constants(nullptr, &mu0, nullptr, nullptr);

    fdtd3d_desc d{};
    d.struct_size = sizeof d;
    d.Nx = c.Nx; d.Ny = c.Ny; d.Nz = c.Nz;
    for (int q = 0; q < 6; ++q) d.n_pml[q] = c.n_pml[q];
    d.k_Jx = c.k_Jx; d.n_wf = c.steps; d.precision = c.precision; d.has_mu = 0;
    d.mode = c.fused ? FDTD3D_MODE_FUSED : (c.lean ? FDTD3D_MODE_LEAN : FDTD3D_MODE_FULL);
    d.z_lo = 0; d.z_hi = c.Nz; d.device = -1;
    d.dx = c.dx; d.dy = c.dy; d.dz = c.dz; d.dt = c.dt; d.mu0 = mu0;
    int slabs = c.slabs > 0 ? c.slabs : fdtd_device_count();
    if (slabs < 1) slabs = 1;
    if (slabs > c.Nz) slabs = c.Nz;
    must(fdtd3d_group_create(&d, slabs, nullptr, &m_group), "fdtd3d_group_create");
    if (slabs > 1) std::cout << "[grid] " << slabs << " z-slabs on " << std::min(slabs, std::max(fdtd_device_count(), 1)) << " GPU(s)\n";

    // source, materials, CPML: generated on the host once, handed over, then dropped
    {
        std::vector<double> wf(c.steps), mag(static_cast<size_t>(c.Nx) * (c.Ny + 1));
        must(fdtd3d_default_source(c.Nx, c.Ny, c.steps, c.dx, c.dy, c.dz, c.dt, wf.data(), mag.data()), "default source");
        for (int r = 0; r < slabs; ++r) {          // source and CPML tables are not split: every slab holds them
            must(fdtd3d_set_array(fdtd3d_group_member(m_group, r), FDTD3D_JX_WF, wf.data(), wf.size()), "upload waveform");
            must(fdtd3d_set_array(fdtd3d_group_member(m_group, r), FDTD3D_JX_MAG, mag.data(), mag.size()), "upload source magnitude");
        }
    }
    // setupMaterials (reference FDTD3D.cpp:106-200) runs on the device: same fp64 arithmetic, no O(N^3) host loop
    must(fdtd3d_group_fill_reference_materials(m_group), "reference materials");
    const double h[3] = {c.dx, c.dy, c.dz};
    for (int axis = 0; axis < 3; ++axis) for (int side = 0; side < 2; ++side) {
        const int n = c.n_pml[axis * 2 + side];
        if (n == 0) continue;
        std::vector<double> t(static_cast<size_t>(6) * n);
        must(fdtd_cpml_tables(n, side, h[axis], h[axis], c.dt, t.data()), "CPML tables");
        for (int kind = 0; kind < 6; ++kind)
            for (int r = 0; r < slabs; ++r)
                must(fdtd3d_set_array(fdtd3d_group_member(m_group, r), FDTD3D_TAB + (axis * 2 + side) * 6 + kind, t.data() + static_cast<size_t>(kind) * n, n), "upload CPML table");
    }
    if (c.write_files) m_out = std::make_unique<FieldWriter3D>(c.out_dir, c.binary_vti ? FieldWriter3D::Encoding::appended : FieldWriter3D::Encoding::ascii);
    const size_t cells = static_cast<size_t>(c.Nx) * c.Ny * c.Nz;
    m_a.assign(cells, 0.0);
    m_b.assign(cells, 0.0);
}

void FDTD3D::dump_frame(int step) {
    const Sim3DConfig& c = m_cfg;
    must(fdtd3d_group_observe_frame(m_group, m_a.data(), m_b.data(), m_a.size()), "observe frame");
    if (!m_out) return;
    m_out->write_vti("Ex", m_a, c.Nx, c.Ny, c.Nz, c.dx, c.dy, c.dz, step);
    m_out->write_vti("E_mag", m_b, c.Nx, c.Ny, c.Nz, c.dx, c.dy, c.dz, step);
}

void FDTD3D::run() {
    const Sim3DConfig& c = m_cfg;
    const int half_period = c.n_T0 / 2;
    const double weight = 4.0 / c.n_T0;
    // The reference looks at the fields after every step (FDTD3D.cpp:1085-1108): the RMS window is advanced, a completed
    // window written out, every frame_every-th step dumped.  Here the steps between two of those moments go to the engine
    // in one call, which advances the window on the device by itself (two steps per launch on large grids).  A run of
    // steps ends at the next step whose fields are dumped, and one step after the next step that completes a window at
    // the latest -- the engine keeps a completed window until the next one completes, so the step that follows it may
    // still ride in the same launch.
    const int every = c.frame_every > 0 ? c.frame_every : c.steps + 1;
    for (int n = 0; n < c.steps;) {
        const auto t0 = std::chrono::steady_clock::now();
        const int next_dump = (n + every - 1) / every * every;                                   // first step >= n that is dumped
        const int next_done = (n + half_period) / half_period * half_period - 1;                 // first step >= n that completes a window
        const int last = c.step_by_step ? n : std::min(std::min(next_dump, next_done + 1), c.steps - 1);
        must(fdtd3d_group_step_observed(m_group, n, last - n + 1, weight, half_period), "step");
        must(fdtd3d_group_sync(m_group), "sync");
        m_device_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (next_done <= last) {
            must(fdtd3d_group_observe_rms_window(m_group, m_a.data(), m_a.size()), "RMS download");
            if (m_out) m_out->write_vti("E_mag_rms", m_a, c.Nx, c.Ny, c.Nz, c.dx, c.dy, c.dz, next_done + 1);
        }
        if (last % every == 0) {
            dump_frame(last);
            std::cout << "[n=" << last << "] frame dumped.\n";
        }
        n = last + 1;
        m_step_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    std::cout << "Done. VTI in ./" << c.out_dir << "\n";
}
