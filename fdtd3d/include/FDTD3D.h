// FDTD3D.h -- host driver of the 3-D solver on top of the MI355X engine.
//
// Same public face as the reference class (constructor, destructor, run(); see the reference's
// fdtd3d/include/FDTD3D.h:10-15) so that `FDTD3D sim; sim.run();` keeps working.  Everything the
// reference kept in private std::vector members lives on the device behind fdtd_engine.h; what is
// left on the host is configuration, the once-per-run setup and the output cadence of run().
#pragma once
#include <memory>
#include <string>
#include <vector>

struct fdtd3d_group;
class FieldWriter3D;

struct Sim3DConfig {
    int Nx, Ny, Nz, steps;
    double dx, dy, dz, dt;
    int n_pml[6];          // xn xp yn yp zn zp
    int k_Jx;
    int precision;         // 64 reproduces the reference's files byte for byte; 32 is the fast path
    bool lean;             // keep only D and B on the device
    bool fused;            // one kernel per time step (implies lean storage, two buffer sets)
    int slabs;             // z-slabs, slab r on GPU r modulo the visible count; 0 = one per visible GPU
    int n_T0;              // steps per optical period: RMS window = n_T0/2, weights 4/n_T0
    int frame_every;
    bool write_files;
    bool step_by_step;     // one engine call per step, as the reference's loop is written (for comparison: the engine then cannot take two steps per launch)
    bool binary_vti;       // opt-in: raw appended Float32 VTI instead of the reference's ASCII Float64
    std::string out_dir;
    // The reference's hard-coded problem (FDTD3D.cpp:21-39, :71, FDTD3D.h:54-56).
    static Sim3DConfig reference_default();
    // Overrides from FDTD3D_NX / _NY / _NZ / _STEPS / _NPML / _KJX / _PRECISION / _LEAN / _FUSED / _SLABS / _FRAME_EVERY / _BINARY_VTI /
    // _WRITE / _STEP_BY_STEP / _OUT environment variables (the reference's main() takes no arguments).
    void apply_environment();
};

class FDTD3D {
public:
    FDTD3D();
    explicit FDTD3D(const Sim3DConfig& cfg);
    ~FDTD3D();
    FDTD3D(const FDTD3D&) = delete;
    FDTD3D& operator=(const FDTD3D&) = delete;

    void run();

    const Sim3DConfig& config() const { return m_cfg; }
    double seconds_in_steps() const { return m_step_seconds; }          // time loop incl. downloads and file output
    double seconds_on_device() const { return m_device_seconds; }       // stepping + RMS window on the device only

private:
    void build();
    void dump_frame(int step);

    Sim3DConfig m_cfg;
    fdtd3d_group* m_group = nullptr;   // the slabs of the grid (a single one by default)
    std::unique_ptr<FieldWriter3D> m_out;
    std::vector<double> m_a, m_b;     // download buffers, [Nz][Ny][Nx]
    double m_step_seconds = 0.0, m_device_seconds = 0.0;
};
