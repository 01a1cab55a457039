// FDTD2D.h -- host driver of the 2-D (Ey, Hx, Hz) solver on top of the MI355X engine.  Public face
// as in the reference (constructor, destructor, run(); fdtd2d/include/FDTD2D.h:9-14); fields,
// Lorentz histories and CPML state live on the device behind fdtd_engine.h.
#pragma once
#include <memory>
#include <string>
#include <vector>

struct fdtd2d_engine;
class FieldWriter2D;

struct Sim2DConfig {
    int Nx, Nz, steps;
    double dx, dz, dt;
    int n_pml[4];          // xn xp zn zp
    int i_Jy, k_Jy;
    int slab;              // 1: the reference's Lorentz slab, 0: no dispersive material
    int precision;
    bool fused;            // one kernel per step (Ey, Hx, Hz only stored at dispersive cells)
    bool step_by_step;     // one engine call per step, as the reference's loop is written (for comparison: the engine then cannot take two steps per launch)
    int n_T0;              // steps per period: RMS window n_T0/2, weights 4/n_T0
    int frame_every;
    bool write_files;
    std::string out_dir;
    // The reference's hard-coded problem (FDTD2D.cpp:20-38, :60-65, FDTD2D.h:41).
    static Sim2DConfig reference_default();
    // FDTD2D_NX / _NZ / _STEPS / _NPML / _IJY / _KJY / _SLAB / _PRECISION / _FRAME_EVERY / _FUSED / _STEP_BY_STEP / _WRITE / _OUT
    void apply_environment();
};

class FDTD2D {
public:
    FDTD2D();
    explicit FDTD2D(const Sim2DConfig& cfg);
    ~FDTD2D();
    FDTD2D(const FDTD2D&) = delete;
    FDTD2D& operator=(const FDTD2D&) = delete;

    void run();
    const Sim2DConfig& config() const { return m_cfg; }
    double seconds_in_steps() const { return m_step_seconds; }            // time loop incl. downloads and file output
    double seconds_on_device() const { return m_device_seconds; }         // stepping + RMS window on the device only

private:
    void build();
    void dump_frame(int step);

    Sim2DConfig m_cfg;
    fdtd2d_engine* m_engine = nullptr;
    std::unique_ptr<FieldWriter2D> m_out;
    std::vector<double> m_ey, m_center;
    double m_step_seconds = 0.0, m_device_seconds = 0.0;
};
