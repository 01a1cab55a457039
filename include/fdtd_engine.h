/*
 * fdtd_engine.h -- C ABI of the MI355X (gfx950) FDTD time-stepping engine.
 *
 * The engine replaces the per-step hot path of the two reference solvers:
 *   3-D: FDTD3D::updateE(int) + FDTD3D::updateH()      (fdtd3d/include/FDTD3D.h:190-195,
 *        bodies fdtd3d/src/FDTD3D.cpp:480-1021; called from FDTD3D::run(), :1085-1086)
 *   2-D: FDTD2D::stepOnce(int)                          (fdtd2d/include/FDTD2D.h:88-89,
 *        body fdtd2d/src/FDTD2D.cpp:275-482; called from FDTD2D::run(), :512)
 *
 * In the reference these are private void member functions acting on private
 * std::vector<double> members; there is no error path.  Here the state lives in
 * device memory behind an opaque handle, the caller hands over the setup results
 * (material arrays, CPML tables, source) once, and every call returns an int
 * status (0 = ok) with a thread-local message behind fdtd_last_error().
 *
 * Host arrays crossing this boundary are always double, in the reference's own
 * flat order ("canonical order"): 3-D [i][j][k] with k unit-stride
 * (FDTD3D.h:115-125), 2-D [i][k] with k unit-stride (FDTD2D.h:70-76).  The engine
 * converts to its own device layout (x unit-stride, z slowest, rows padded to
 * 256 elements) and to the compute precision (fp32 or fp64) internally.
 */
#ifndef FDTD_ENGINE_H
#define FDTD_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FDTD_OK 0
#define FDTD_ERR_ARG 1      /* bad descriptor / id / size                      */
#define FDTD_ERR_HIP 2      /* a HIP runtime call failed (message has details) */
#define FDTD_ERR_STATE 3    /* call not valid for this handle                  */
#define FDTD_ERR_HOST 4     /* host side failed (out of memory while staging)  */

/* Message for the last failing call on this thread ("" if none). */
const char* fdtd_last_error(void);
/* Library build identification ("fdtd_engine gfx950 ..."). */
const char* fdtd_engine_version(void);
/* Number of HIP devices this process can use (0 if there is none). */
int fdtd_device_count(void);
/* fp32 division by a constant (eps0, mu0 in the 2-D two-step kernel): 1 if the three-instruction form x*r corrected by
 * two fused multiply-adds has been proven, by running every fp32 numerator of its class (|x| within 2^-80..2^80 scaled
 * by c, or +0) through both on the current device, to give the IEEE quotient bit for bit -- the kernels then use it for
 * waves whose numerators are all in the class and the IEEE division otherwise; 0 if not (the kernels always divide);
 * -1 on error.  The environment variable FDTD_CONST_DIV=0 turns the short form off. */
int fdtd_const_division_proven(double c);

/* ------------------------------------------------------------------ 3-D -- */

/* Array ids, 3-D.  Shapes (canonical order, k fastest), from FDTD3D.h:31-51:
 *   Dx,Ex,eps_x [Nx][Ny+1][Nz+1]   Dy,Ey,eps_y [Nx+1][Ny][Nz+1]   Dz,Ez,eps_z [Nx+1][Ny+1][Nz]
 *   Bx,Hx,mu_x  [Nx+1][Ny][Nz]     By,Hy,mu_y  [Nx][Ny+1][Nz]     Bz,Hz,mu_z  [Nx][Ny][Nz+1] */
enum {
    FDTD3D_DX = 0, FDTD3D_DY = 1, FDTD3D_DZ = 2,
    FDTD3D_EX = 3, FDTD3D_EY = 4, FDTD3D_EZ = 5,
    FDTD3D_BX = 6, FDTD3D_BY = 7, FDTD3D_BZ = 8,
    FDTD3D_HX = 9, FDTD3D_HY = 10, FDTD3D_HZ = 11,
    FDTD3D_EPS_X = 12, FDTD3D_EPS_Y = 13, FDTD3D_EPS_Z = 14,
    FDTD3D_MU_X = 15, FDTD3D_MU_Y = 16, FDTD3D_MU_Z = 17,   /* extension: reference has mu0 only */
    /* CPML auxiliary fields (FDTD3D.h:80-103).  id = base + (comp*3 + axis)*2 + side,
     * comp = component of D (resp. B) the term updates (0=x,1=y,2=z), axis = derivative
     * axis (!= comp), side 0 = low face, 1 = high face.  Canonical shape = the shape of
     * that D (resp. B) component with the extent along `axis` replaced by the layer
     * count of that face; the entry for cell (i,j,k) in layer l sits at the cell's own
     * transverse indices (the reference's per-array index shifts, FDTD3D.h:136-182, are
     * undone).  Entries no sweep touches are zero. */
    FDTD3D_PHI_E = 100,
    FDTD3D_PHI_M = 200,
    /* CPML 1-D tables (FDTD3D.h:58-78).  id = base + (axis*2 + side)*6 + kind,
     * kind: 0 b_e, 1 a_e, 2 C_e (=dt/d/kappa_e), 3 b_m, 4 a_m, 5 C_m; length = layers. */
    FDTD3D_TAB = 300,
    FDTD3D_JX_MAG = 400,      /* [Nx][Ny+1], FDTD3D.h:107 */
    FDTD3D_JX_WF = 401        /* [n_wf],     FDTD3D.h:108 */
};

/* Storage modes. */
#define FDTD3D_MODE_FULL 0   /* D,E,B,H all resident (the reference's formulation)            */
#define FDTD3D_MODE_LEAN 1   /* only D,B resident; E=D/eps, H=B/mu recomputed where consumed */
#define FDTD3D_MODE_FUSED 2  /* lean storage plus second D, B and D-side CPML buffers: one kernel
                                per time step (fdtd3d_step / fdtd3d_fused), both halves fused */

typedef struct fdtd3d_desc {
    int32_t struct_size;      /* sizeof(fdtd3d_desc), for forward compatibility             */
    int32_t Nx, Ny, Nz;       /* global cell counts (FDTD3D.h:25)                            */
    int32_t n_pml[6];         /* xn, xp, yn, yp, zn, zp (FDTD3D.h:54-56)                    */
    int32_t k_Jx;             /* source plane (FDTD3D.h:106)                                 */
    int32_t n_wf;             /* waveform length (m_steps)                                   */
    int32_t precision;        /* 32 or 64.  64 is the mode that matches the reference: bit for bit, over its whole
                                 run lengths.  32 equals an fp32 restatement of the reference's loops bit for bit and
                                 stays within 1e-6 relative L2 of the fp64 reference for a limited number of steps
                                 (measured: about 700 on the 3-D default problem, where the drift grows as
                                 4e-8*sqrt(steps); about 20 for H in a lossless cavity, 600 for E; about 150 on the
                                 2-D default problem, whose undamped Lorentz recursion makes it grow quadratically) */
    int32_t has_mu;           /* 0: H = B/mu0 (reference), 1: H = B/mu_{x,y,z}               */
    int32_t mode;             /* FDTD3D_MODE_*                                               */
    int32_t z_lo, z_hi;       /* owned cell layers [z_lo, z_hi); 0,Nz = whole grid           */
    int32_t device;           /* HIP device ordinal, -1 = the calling thread's current one   */
    double dx, dy, dz, dt;    /* FDTD3D.h:26-28                                              */
    double mu0;               /* FDTD3D.h:20                                                 */
} fdtd3d_desc;

typedef struct fdtd3d_engine* fdtd3d_handle;

/* Allocate device state for one z-slab (fields and CPML auxiliaries zero, eps/mu = 1,
 * tables/source empty).  Replaces the allocations of FDTD3D::setupGrid()/setupPML()
 * (FDTD3D.cpp:41-63, :419-462).  The handle remembers its device: every later call makes that
 * device current for its own duration and puts the caller's device back before it returns, so
 * handles on several devices may be driven from one thread.  Streams passed to a call must
 * belong to the handle's device. */
int fdtd3d_create(const fdtd3d_desc* desc, fdtd3d_handle* out);
int fdtd3d_destroy(fdtd3d_handle h);

/* Number of doubles of the GLOBAL canonical array `id` (0 for an unknown id). */
size_t fdtd3d_array_size(fdtd3d_handle h, int id);
/* Upload from / download into a GLOBAL canonical array.  A slab engine takes the part
 * it stores (owned layers plus its ghost planes) and, on download, writes only the
 * layers it owns, leaving the rest of `host` untouched.  In LEAN mode E and H are
 * computed on the fly for download and cannot be uploaded. */
int fdtd3d_set_array(fdtd3d_handle h, int id, const double* host, size_t n);
int fdtd3d_get_array(fdtd3d_handle h, int id, double* host, size_t n);

/* Device-side synthetic materials for grids too large for host arrays:
 * eps_c(cell) = eps_lo + (eps_hi-eps_lo)*u, u = fdtd_hash_unit(seed, c, canonical index)
 * and likewise mu (only if has_mu).  Same function as oracle twin3d_fill_materials. */
int fdtd3d_fill_materials(fdtd3d_handle h, uint64_t seed, double eps_lo, double eps_hi,
                          double mu_lo, double mu_hi);

/* The same counter hash into any resident field array (ids 0..17), values in [lo, hi): a state without
 * zeros for measurements (is the step time independent of the data?) and tests; ghost planes included. */
int fdtd3d_fill_array(fdtd3d_handle h, int id, uint64_t seed, double lo, double hi);

/* FDTD3D::setupMaterials (FDTD3D.cpp:106-200) evaluated on the device for the layers this engine
 * stores: fused silica (n = 1.47) with a centred air sphere of radius two wavelengths, 3 x 3
 * sub-sampled -- fp64, operation for operation as fdtd3d_default_materials() on the host (the two
 * agree bit for bit), without the O(N^3) single-threaded host loop and the host arrays. */
int fdtd3d_fill_reference_materials(fdtd3d_handle h);

/* Advance `count` whole steps n0, n0+1, ...: per step E-half (curl+CPML, source, E=D/eps;
 * FDTD3D::updateE) then H-half (FDTD3D::updateH).  Asynchronous on `stream`
 * (a hipStream_t, NULL = default stream).  Only valid when the slab is the whole grid. */
int fdtd3d_step(fdtd3d_handle h, int n0, int count, void* stream);

/* Steps n0 .. n0 + 2*pairs - 1 with TWO steps per launch (updateE, updateH, updateE, updateH of
 * FDTD3D.cpp:480-1021 for a cell before its neighbours' second step is due): D, B, eps and mu cross
 * device memory once per two steps; the results are those of 2*pairs calls of the one-step kernel, bit
 * for bit, CPML state included.  Restricted to FUSED mode; on a slab see the note on overlapping slabs
 * below.  Anything else is refused with FDTD_ERR_STATE.  Registered probes are sampled after BOTH steps
 * of a launch on an engine that holds the whole grid: the values after the first step, which never
 * reach memory, are formed before the launch by one-plane launches of the one-step kernel over the
 * planes that hold probes (a slab engine with probes is refused).  fdtd3d_step uses these launches by
 * itself for as many of its steps as it can when the engine holds the whole grid and the grid is large
 * enough for them to pay (384^3 cells, 448^3 with CPML layers; the environment variable FDTD_PAIR=0/1 overrides).
 * fdtd3d_pair_shape sets the planes per workgroup (default: chosen from the grid, at most 128) and the
 * waves per workgroup (6..16, default 16; three of them are halo rows); 0 keeps. */
int fdtd3d_step_pairs(fdtd3d_handle h, int n0, int pairs, void* stream);
int fdtd3d_pair_shape(fdtd3d_handle h, int chunk, int rows);
/* Two-step launches on a SLAB (z_lo > 0 or z_hi < Nz) are exact only two planes away from a cut: the kernel
 * recomputes its halos from what the slab stores and the slab's single ghost plane cannot feed the second step
 * of the plane next to it.  A slab driver therefore lets neighbouring slabs OVERLAP by one layer either side of
 * the cut (rank r stores [z_lo-1, z_hi+1) and answers for [z_lo, z_hi)), calls fdtd3d_step_pairs(h, n, 1, ..)
 * and then overwrites the two planes beyond each cut -- D, B and the x- and y-face CPML state, both families --
 * with the neighbour's exact ones (fdtd3d_planes gives the device ranges; fdtd_amd/slab.py: PairSlabStepper).
 * Cuts must stay clear of the z layers (the z-face state is not exchanged). */
/* Device range of `count` consecutive stored planes, starting at GLOBAL plane k, of array `id` (D or B
 * component, or an x- or y-face CPML array) in the buffers that are current now -- fetch again after every
 * step, the buffers change roles.  Pointer arithmetic only. */
int fdtd3d_planes(fdtd3d_handle h, int id, int k, int count, void** p, size_t* bytes);
/* Steps n and n+1 in parts, so that a slab driver can have the planes next to its cuts on their way while
 * the bulk is computed: FDTD3D_PART_EDGE = the three lowest and the three highest owned planes (two small
 * launches; the buffers change roles here, as with fdtd3d_fused, so fdtd3d_planes already refers to the new
 * values), then FDTD3D_PART_INTERIOR = the rest.  FDTD3D_PART_ALL = fdtd3d_step_pairs(h, n, 1, stream). */
int fdtd3d_pair_part(fdtd3d_handle h, int n, int part, void* stream);
/* Both parts of a pair must be launched on the same stream (all launches of an engine share one piece of scratch
 * memory); between EDGE and INTERIOR every other stepping call on the engine is refused with FDTD_ERR_STATE.
 * 1 if a two-step launch of `workgroups` workgroups of `waves_per_group` waves stays within the scratch memory the
 * kernel can address (it forms a wave's offset in 32 bits: 24 row segments per wave, below 2^32 elements in all), else
 * 0; launches beyond it are refused with FDTD_ERR_STATE, and fdtd3d_step then takes single steps.  Needs no device. */
int fdtd3d_pair_scratch_fits(long long workgroups, int waves_per_group, int precision_bits);
/* out = { 1 if fdtd3d_step takes its steps two per launch on this engine, planes per workgroup,
 *         waves per workgroup, elements per lane } of the two-step kernel. */
int fdtd3d_pair_info(fdtd3d_handle h, int out[4]);
/* 1 if two-step launches can run on this engine at all, whole grid or slab (FUSED storage, CPML state and scratch
 * memory within what the kernel indexes with 32 bits), else 0. */
int fdtd3d_pair_capable(fdtd3d_handle h);
/* out = { x tiles per row, lanes of tile 0 that hold the first columns of a row when that tile holds both ends
 *         of it (0: the tiles lie in a row), first column of its other lanes, tiles per run handed to one XCD
 *         (0: tiles are taken as dealt) } of the two-step kernel's next launch.  No counterpart in the reference. */
int fdtd3d_pair_layout(fdtd3d_handle h, int out[4]);

/* Half steps over a sub-range of the owned layers, for slab drivers that overlap the
 * ghost exchange with the interior update.  part: */
#define FDTD3D_PART_ALL 0
#define FDTD3D_PART_EDGE 1       /* E-half: lowest owned layer; H-half: highest owned layer */
#define FDTD3D_PART_INTERIOR 2   /* the rest                                                 */
#define FDTD3D_PART_TOP 3        /* fdtd3d_fused only: the highest owned planes               */
int fdtd3d_half_e(fdtd3d_handle h, int n, int part, void* stream);
int fdtd3d_half_h(fdtd3d_handle h, int part, void* stream);

/* FUSED mode: both halves of step n in one kernel (updateE then updateH of FDTD3D.cpp:480-1021,
 * cell for cell in the same operation order), reading the current D, B and D-side CPML buffers
 * and writing the other set; the two sets change roles when the first part of a step is launched.
 *   PART_ALL       the whole slab.
 *   PART_EDGE      the lowest plane (FDTD_EDGE_PLANES in the environment: more), so that D of layer z_lo can travel down early.
 *   PART_TOP       optional, after EDGE: the highest plane (or planes, likewise), so that the top layer can be
 *                  finished (fdtd3d_fused_top) and its B sent up while the bulk is still running.
 *   PART_INTERIOR  everything the step's earlier parts left; ends the step.
 * A slab below another one leaves B of its top layer z_hi-1 unchanged: once the upper
 * neighbour's D has arrived in the high ghost plane, fdtd3d_fused_top() finishes it with the
 * single-plane H kernel (a no-op on the slab that holds k = Nz).
 * fdtd3d_halo() always refers to the current buffers (fdtd3d_buffer_parity() tells which set,
 * 0 or 1, that is), so pointers must be fetched again after a step or cached per parity. */
int fdtd3d_fused(fdtd3d_handle h, int n, int part, void* stream);
int fdtd3d_fused_top(fdtd3d_handle h, void* stream);
int fdtd3d_buffer_parity(fdtd3d_handle h);

/* Ghost-plane exchange buffers (device pointers into the engine's own arrays; each is
 * one contiguous x-y plane of `bytes` bytes).  which: */
#define FDTD3D_HALO_SEND_UP 0    /* tangential H (FULL) / B (LEAN), top owned layer -> rank+1 */
#define FDTD3D_HALO_RECV_LO 1    /* ghost layer z_lo-1                             <- rank-1 */
#define FDTD3D_HALO_SEND_DOWN 2  /* tangential E (FULL) / D (LEAN), layer z_lo     -> rank-1 */
#define FDTD3D_HALO_RECV_HI 3    /* ghost layer z_hi                               <- rank+1 */
/* comp 0 = x component, 1 = y component. */
int fdtd3d_halo(fdtd3d_handle h, int which, int comp, void** dev_ptr, size_t* bytes);
/* The same for buffer set `parity` (0 or 1) of a FUSED engine, whichever is current. */
int fdtd3d_halo_at(fdtd3d_handle h, int parity, int which, int comp, void** dev_ptr, size_t* bytes);

/* Sum over owned cells of the squares of array `id`, accumulated in double on device
 * (diagnostic / checksum). */
int fdtd3d_sumsq(fdtd3d_handle h, int id, double* out, void* stream);
/* The same two reductions over the owned planes inside the GLOBAL range [k_lo, k_hi) only (either result
 * pointer may be NULL): for slabs that store more layers than they answer for. */
int fdtd3d_reduce_range(fdtd3d_handle h, int id, int k_lo, int k_hi, double* sumsq, uint64_t* checksum, void* stream);

/* Order-independent 64-bit checksum over owned cells of array `id` (value bits mixed with the
 * canonical cell index, summed modulo 2^64; zeros of either sign contribute nothing), for
 * bit-exact comparisons at sizes too large to download. */
int fdtd3d_checksum(fdtd3d_handle h, int id, uint64_t* out, void* stream);

/* fdtd3d_sumsq() and fdtd3d_checksum() for up to FDTD3D_REDUCE_STATIC_MAX field arrays in one call that
 * allocates nothing.  No counterpart in the reference: it is for callers that cannot allocate device
 * memory -- tight serving loops, harnesses that forbid hipMalloc.  The partial sums of the reduction
 * kernel live in a file-scope device array of the library (131,456 bytes on every GPU the library is
 * used on, held until the process ends) instead of memory from hipMalloc; the call makes no
 * hipMalloc/hipFree and uses no allocating host container.  ids[q] in 0..17 must be resident in the
 * engine's storage mode; sumsq[q] / checksum[q] receive what fdtd3d_sumsq / fdtd3d_checksum return for
 * ids[q], bit for bit (either output may be NULL).  More than FDTD3D_REDUCE_STATIC_MAX arrays are
 * refused with FDTD_ERR_ARG.  All calls on one GPU share that GPU's array: calls that overlap in time
 * on one GPU (from several threads, handles or streams) race.  Synchronises `stream` before returning. */
#define FDTD3D_REDUCE_STATIC_MAX 16
int fdtd3d_reduce_static(fdtd3d_handle h, const int* ids, int n_ids, double* sumsq, uint64_t* checksum, void* stream);

/* Cell-centre observables, computed on the device so that the host loop of FDTD3D::run()
 * (FDTD3D.cpp:1068-1108) never downloads the fields.  Arrays are [Nz][Ny][Nx] with x fastest, the
 * order run()/dumpFrame() hand to the VTI writer (FDTD3D.cpp:1028-1030); a slab engine fills the
 * layers it owns.
 *   accumulate: acc(cell) (+)= (1/16)(sum4 Ex)^2+... * weight, acc zeroed first if reset != 0
 *               (FDTD3D.cpp:1074-1097; the reference passes weight = 4.0 / n_T0)
 *   rms:        host = sqrt(acc)                                        (FDTD3D.cpp:1099-1100)
 *   frame:      ex_center = (1/4) sum4 Ex, e_mag = sqrt(ax^2+ay^2+az^2) (FDTD3D.cpp:1035-1051) */
int fdtd3d_observe_accumulate(fdtd3d_handle h, double weight, int reset, void* stream);
int fdtd3d_observe_rms(fdtd3d_handle h, double* host, size_t n);
/* The loop body of FDTD3D::run() (FDTD3D.cpp:1085-1097) for `count` steps from n0: step, then accumulate with
 * `weight`; a new window opens (the accumulator is zeroed first) at every step n with n % window == 0, and the
 * window that a step n completes ((n + 1) % window == 0; the reference writes it out there, FDTD3D.cpp:1098-1103)
 * is kept and can be fetched with fdtd3d_observe_rms_window() until the next one completes.  Where fdtd3d_step
 * would take two steps per launch, the launch advances the window by both of its steps from the E values it has
 * on chip -- same additions in the same order, so the window equals, bit for bit, what step + accumulate per step
 * give; a window completed by the first step of such a launch is kept on the way.  Whole-grid engines only.
 *   keep_window: what fdtd3d_step_observed does when a step completes a window, for callers that step and
 *                accumulate themselves (slab drivers): the accumulator as it stands becomes the kept window
 *                (no copy: the two arrays change roles), and the next accumulate must open a new window.
 *   rms_window:  host = sqrt(kept window), laid out like fdtd3d_observe_rms. */
/* For drivers of overlapping slabs (see fdtd3d_pair_part):
 *   set_range:          the layers [k_lo, k_hi) this engine reports observables for (inside what it stores; default: all
 *                       it stores).  Before the first download.
 *   pair_part_observed: fdtd3d_pair_part with the window advanced by both steps of the launch (`window` as in
 *                       fdtd3d_step_observed).  defer_top != 0 (a slab with a neighbour above): the second step's
 *                       contribution of the top reported layer is left out -- it needs E one layer further up, which is
 *                       exact only once the planes beyond the cut have arrived.
 *   finish_pair:        after those planes have arrived: adds the deferred contribution (one-layer launch of the
 *                       observe kernel) and, if the launch's second step completed a window, keeps it. */
int fdtd3d_observe_set_range(fdtd3d_handle h, int k_lo, int k_hi);
int fdtd3d_pair_part_observed(fdtd3d_handle h, int n, int part, double weight, int window, int defer_top, void* stream);
int fdtd3d_observe_finish_pair(fdtd3d_handle h, int n, double weight, int window, int defer_top, void* stream);
int fdtd3d_step_observed(fdtd3d_handle h, int n0, int count, double weight, int window, void* stream);
int fdtd3d_observe_keep_window(fdtd3d_handle h);
int fdtd3d_observe_rms_window(fdtd3d_handle h, double* host, size_t n);
int fdtd3d_observe_frame(fdtd3d_handle h, double* ex_center, double* e_mag, size_t n, void* stream);

/* Probe time series (no counterpart in the reference, which has no probes; the value sampled is the
 * one a read of the reference's vector at that cell would give after the step).  Up to 64 cells of
 * field arrays (ids 0..17), given by canonical (i, j, k); a slab engine only accepts cells of layers
 * it owns.  fdtd3d_step records one sample of every probe after each step; slab drivers call
 * fdtd3d_probe_record() themselves at the end of a step.  Samples are kept on the device;
 * fdtd3d_probe_read() synchronises and returns them sample-major, out[sample * n_probes + probe]
 * (out may be NULL to query the counts).  Probes are added before the first sample or after
 * fdtd3d_probe_reset(h, 1); reset(h, 0) only drops the samples. */
int fdtd3d_probe_add(fdtd3d_handle h, int id, int i, int j, int k);
int fdtd3d_probe_record(fdtd3d_handle h, void* stream);
int fdtd3d_probe_read(fdtd3d_handle h, double* out, size_t cap, int* n_samples, int* n_probes);
int fdtd3d_probe_reset(fdtd3d_handle h, int drop_probes);

/* Placement of large engines (2 GiB of device state or more).  The step kernels stream from all field arrays at once,
 * and the rate depends on how far apart those arrays lie in physical device memory: adjacent is slowest, about 10 GiB of
 * other memory between two arrays fastest (1024^3, two steps per launch: 11.85 ms per step with 4 GiB between the arrays,
 * 11.12 with 10).  An engine that holds a whole grid therefore backs its arrays through the virtual-memory API with
 * physical chunks created in order, and creates and releases up to 10 GiB of further chunks in front of every field array:
 * the arrays end up that far apart physically while the process only holds their own memory.  A slab engine reaches the
 * same layout with ordinary allocations -- every array a hipMalloc of its own, each large one behind a throw-away
 * allocation of the gap, all of which are freed once the arrays exist -- because its planes travel between devices.  Both
 * take less where less is free, and fall back to one plain allocation when no useful gap is left (a slab engine whose
 * gaps leave no room for an array gives them back at once and goes on without).  While an engine is being created it
 * transiently holds that extra memory: two engines created at the same moment on one device can fail for lack of it.
 * fdtd3d_spread_gap_gib() reports the smallest gap achieved (0: the arrays lie side by side), fdtd3d_spread_gaps() the
 * gap in front of every large array but the first, in allocation order (returns their number, copies up to `cap`).
 * Environment: FDTD_SPREAD=0 off, FDTD_SPREAD=1 the virtual-memory way for every engine, FDTD_SPREAD_GAP_MB the gap.
 * Memory backed through the virtual-memory API must be opened to another device before that device can copy to or from
 * it: fdtd3d_grant_peer_access() (the slab group does it for neighbouring slabs). */
double fdtd3d_spread_gap_gib(fdtd3d_handle h);
int fdtd3d_spread_gaps(fdtd3d_handle h, double* gib, int cap);
int fdtd3d_grant_peer_access(fdtd3d_handle h, int peer_device);

/* Bytes of device memory held by the handle. */
size_t fdtd3d_device_bytes(fdtd3d_handle h);

/* Launch-shape knobs (no effect on results): planes marched per workgroup (1..4096); tile order over
 * the 8 XCDs (0: as dealt, 1: each XCD a contiguous range of tiles, 2..65535: runs of that many
 * consecutive tiles per XCD, 65536: automatic, the default -- for the fused kernel runs of one row
 * group's x tiles); rows (= waves) per workgroup (1..4 for the half-step kernels; 2..16 for the
 * fused kernel, whose last wave recomputes the first row of the next tile); a value outside its
 * range keeps the current setting.  The environment variables FDTD_Z_CHUNK / FDTD_XCD_REMAP / FDTD_ROWS_PER_GROUP
 * set them at creation; FDTD_FUSED_VEC (elements per lane of the fused kernel: 4 or 2 in fp32, 2 or 1
 * in fp64) can only be chosen there. */
int fdtd3d_tune(fdtd3d_handle h, int chunk, int remap, int rows);
/* The launch shape in effect: out = { planes per workgroup, rows per workgroup, tile order (runs of that
 * many tiles per XCD; 0 as dealt, 1 contiguous), elements per lane, planes of the EDGE / TOP launches of
 * a fused slab step, row pitch in elements }.  Measurements record it next to their numbers. */
int fdtd3d_launch_shape(fdtd3d_handle h, int out[6]);

/* Timing hook for the dominant kernels: if enabled, every E-half / H-half launch is
 * bracketed by HIP events on its stream; fdtd3d_kernel_time() synchronises and returns
 * the launch count and the summed device time (ms) per kernel kind (0 = E, 1 = H, 2 = fused step, 3 = two steps per launch). */
int fdtd3d_kernel_timing(fdtd3d_handle h, int enable);
int fdtd3d_kernel_time(fdtd3d_handle h, int kind, int* launches, double* total_ms);
/* The launches timed since the last fdtd3d_kernel_time() call of their kind, in launch order: kind as
 * above, the FDTD3D_PART_* the launch covered, its start relative to the first launch's start and its
 * duration (ms).  The gap between one launch's end and the next one's start on the same stream is the
 * time the stream waited (for a ghost plane, on a slab).  At most `cap` entries are written;
 * *n_launches receives how many there are.  Synchronises; leaves the records in place. */
int fdtd3d_kernel_timeline(fdtd3d_handle h, int cap, int* kinds, int* parts, double* start_ms, double* dur_ms, int* n_launches);

/* ---- Several slabs driven from one host thread ------------------------------------------
 * FDTD3D::run() steps its whole grid from one thread (FDTD3D.cpp:1062-1111).  A group is that
 * loop for a grid cut into `n_slabs` balanced z-slabs, slab r on device devices[r] (NULL: r modulo
 * the visible device count): fdtd3d_group_step() launches every slab's kernels on streams of its
 * own device, edge planes first, and copies the ghost planes device to device (peer copies over
 * xGMI between different devices) while the bulk of each slab is computed.  `desc` describes the
 * whole grid (z_lo = 0, z_hi = Nz; `device` is ignored); all three storage modes work.  Results
 * equal those of one engine holding the whole grid, bit for bit.
 * The set/get/fill/observe calls below address the whole problem with the same global host arrays
 * as the single-engine calls; fdtd3d_group_member() hands out a slab's own handle for anything
 * else (CPML tables and the source go to every member; probes to the member that owns the cell).
 * Calls return once the work is queued; fdtd3d_group_sync() waits for it (downloads do so
 * themselves). */
typedef struct fdtd3d_group* fdtd3d_group_handle;
int fdtd3d_group_create(const fdtd3d_desc* desc, int n_slabs, const int* devices, fdtd3d_group_handle* out);
int fdtd3d_group_destroy(fdtd3d_group_handle g);
int fdtd3d_group_slabs(fdtd3d_group_handle g);
/* FUSED groups of slabs large enough for it (the sizes from which fdtd3d_step pairs its steps, counted per slab;
 * FDTD_PAIR=0/1 in the environment overrides) take TWO steps per launch as well: their slabs then overlap by one
 * layer either side of every cut (slab r stores [lo-1, hi+1) and answers for [lo, hi), see the note on overlapping
 * slabs above), every slab's launch is cut into the planes next to its cuts and the bulk, and after the first part
 * the two planes beyond each cut are copied over from the neighbour while the bulk runs.  Needs every slab two
 * layers thick and every cut clear of the z-face CPML layers; otherwise, and for groups with probes (sampled after
 * every single step), the one-step choreography runs.  fdtd3d_group_pairs() tells (1/0); fdtd3d_group_bounds() gives
 * the layers [lo, hi) slab r answers for (what its handle stores may be one layer more either side). */
int fdtd3d_group_pairs(fdtd3d_group_handle g);
/* How the planes travel between the slabs of groups created from now on: 0 = device-to-device copies issued by the
 * sender's copy stream (hipMemcpyPeerAsync between different devices; the default), 1 = RCCL: one communicator per
 * distinct device (ncclCommInitAll), ncclSend on the sender's copy stream and ncclRecv on a stream of the receiving
 * slab's device, one ncclGroupStart/End per neighbour exchange; slabs that share a device share its rank and exchange
 * with themselves.  -1 = let FDTD_GROUP_TRANSPORT in the environment decide ("rccl"; anything else: copies).  The RCCL
 * library is looked for at run time, only when asked for (a copy the process has already loaded, else the ROCm
 * installation's); group creation fails with FDTD_ERR_STATE if it cannot be found.  Same results either way.
 * fdtd3d_group_transport() tells what a group uses. */
void fdtd3d_group_default_transport(int transport);
int fdtd3d_group_transport(fdtd3d_group_handle g);
int fdtd3d_group_bounds(fdtd3d_group_handle g, int r, int* lo, int* hi);
/* Every event of a group is recorded through a check that event and stream were created for the same
 * slab (HIP refuses to record an event on a stream of another device; with all slabs on one device
 * it would not notice).  A mismatch makes fdtd3d_group_step() fail with FDTD_ERR_STATE. */
fdtd3d_handle fdtd3d_group_member(fdtd3d_group_handle g, int r);
int fdtd3d_group_step(fdtd3d_group_handle g, int n0, int count);
int fdtd3d_group_sync(fdtd3d_group_handle g);
int fdtd3d_group_set_array(fdtd3d_group_handle g, int id, const double* host, size_t n);
int fdtd3d_group_get_array(fdtd3d_group_handle g, int id, double* host, size_t n);
int fdtd3d_group_fill_materials(fdtd3d_group_handle g, uint64_t seed, double eps_lo, double eps_hi, double mu_lo, double mu_hi);
int fdtd3d_group_fill_reference_materials(fdtd3d_group_handle g);
int fdtd3d_group_observe_accumulate(fdtd3d_group_handle g, double weight, int reset);
int fdtd3d_group_observe_rms(fdtd3d_group_handle g, double* host, size_t n);
int fdtd3d_group_step_observed(fdtd3d_group_handle g, int n0, int count, double weight, int window);
int fdtd3d_group_observe_rms_window(fdtd3d_group_handle g, double* host, size_t n);
int fdtd3d_group_observe_frame(fdtd3d_group_handle g, double* ex_center, double* e_mag, size_t n);

/* ------------------------------------------------------------------ 2-D -- */

/* Array ids, 2-D.  Shapes (canonical order, k fastest), FDTD2D.h:33-38:
 *   Dy,Ey,SEy (+ _n1,_n2) [Nx+1][Nz+1];  Bx,Hx,SHx (+..) [Nx+1][Nz];
 *   Bz,Hz,SHz (+..) [Nx][Nz+1];  C1,C2,C3 [Nx][Nz]. */
enum {
    FDTD2D_DY = 0, FDTD2D_EY = 1, FDTD2D_EY_N1 = 2, FDTD2D_EY_N2 = 3,
    FDTD2D_SEY = 4, FDTD2D_SEY_N1 = 5, FDTD2D_SEY_N2 = 6,
    FDTD2D_BX = 7, FDTD2D_HX = 8, FDTD2D_HX_N1 = 9, FDTD2D_HX_N2 = 10,
    FDTD2D_SHX = 11, FDTD2D_SHX_N1 = 12, FDTD2D_SHX_N2 = 13,
    FDTD2D_BZ = 14, FDTD2D_HZ = 15, FDTD2D_HZ_N1 = 16, FDTD2D_HZ_N2 = 17,
    FDTD2D_SHZ = 18, FDTD2D_SHZ_N1 = 19, FDTD2D_SHZ_N2 = 20,
    FDTD2D_C1 = 21, FDTD2D_C2 = 22, FDTD2D_C3 = 23,
    /* CPML auxiliaries (FDTD2D.h:53-59), canonical shape = shape of the updated field
     * with the derivative axis replaced by the layer count (index shifts undone):
     * 100 ey_zn 101 ey_zp 102 ey_xn 103 ey_xp 104 mx_zn 105 mx_zp 106 mz_xn 107 mz_xp */
    FDTD2D_PHI = 100,
    /* tables: id = base + (axis*2+side)*6 + kind, axis 0 = x, 1 = z; kinds as 3-D */
    FDTD2D_TAB = 300,
    FDTD2D_JY = 400           /* [n_wf], FDTD2D.h:63 */
};

typedef struct fdtd2d_desc {
    int32_t struct_size;
    int32_t Nx, Nz;           /* FDTD2D.h:23 */
    int32_t n_pml[4];         /* xn, xp, zn, zp (FDTD2D.h:41) */
    int32_t i_Jy, k_Jy;       /* source cell (FDTD2D.h:62) */
    int32_t n_wf;
    int32_t precision;        /* 32 or 64; 64 matches the reference bit for bit, 32 holds 1e-6 for about 150 steps
                                 on the default problem (see fdtd3d_desc) */
    int32_t device;
    double dx, dz, dt;
    double mu0, eps0;         /* FDTD2D.h:19-20 */
    int32_t mode;             /* FDTD2D_MODE_* */
    int32_t reserved;
} fdtd2d_desc;

/* 2-D storage / stepping modes.  FULL: every array of the reference resident, two kernels per step.
 * FUSED: one kernel per step; Ey, Hx, Hz (and their histories and polarisation sums) are only kept in
 * memory at cells with a non-zero Lorentz coefficient -- elsewhere the field is Dy/eps0 (B/mu0), formed
 * on the fly, and downloads of the current Ey, Hx, Hz derive it the same way; history samples and
 * polarisation sums downloaded from such cells are meaningless (the recursion never reads them there).
 * Upload the coefficient arrays before stepping. */
#define FDTD2D_MODE_FULL 0
#define FDTD2D_MODE_FUSED 1

typedef struct fdtd2d_engine* fdtd2d_handle;

int fdtd2d_create(const fdtd2d_desc* desc, fdtd2d_handle* out);
int fdtd2d_destroy(fdtd2d_handle h);
size_t fdtd2d_array_size(fdtd2d_handle h, int id);
int fdtd2d_set_array(fdtd2d_handle h, int id, const double* host, size_t n);
int fdtd2d_get_array(fdtd2d_handle h, int id, double* host, size_t n);
/* Advance `count` steps n0.. (FDTD2D::stepOnce).  Asynchronous on `stream`.  A FUSED engine without probes takes its
 * steps two per launch (one pass over Dy, Bx, Bz per two steps; an odd step at the end is a launch of the one-step
 * kernel) on grids of at least 4 Mi cells; the environment variable FDTD2D_PAIR=0/1 decides otherwise, and
 * FDTD_Z_CHUNK2D_PAIR sets the rows a wave of that kernel owns (default 8; read when the engine is created).  The result is
 * the same bit for bit either way.  With a probe registered every step is a launch of its own: a probe is sampled
 * after every step. */
int fdtd2d_step(fdtd2d_handle h, int n0, int count, void* stream);
int fdtd2d_sumsq(fdtd2d_handle h, int id, double* out, void* stream);
/* RMS window of Ey on the device (FDTD2D::run, FDTD2D.cpp:515-522): acc(i,k) (+)= Ey^2 * weight for
 * i < Nx, k < Nz; rms writes sqrt(acc) as [Nx][Nz] with k fastest, the order of the CSV writer. */
int fdtd2d_observe_accumulate(fdtd2d_handle h, double weight, int reset, void* stream);
int fdtd2d_observe_rms(fdtd2d_handle h, double* host, size_t n);
/* The body of the reference's time loop for a run of steps (FDTD2D::run, FDTD2D.cpp:510-522): for n = n0 .. n0+count-1
 * step, then acc (+)= Ey^2 * weight with a new window opened when n % window == 0; a window that completes
 * ((n + 1) % window == 0) is kept until the next one completes and read with fdtd2d_observe_rms_window, the one being
 * accumulated with fdtd2d_observe_rms.  Steps that go two per launch (see fdtd2d_step) advance the window inside the
 * launch, from the Ey values the kernel has in registers.  Equal, bit for bit, to fdtd2d_step(n, 1) followed by
 * fdtd2d_observe_accumulate(weight, n % window == 0) for every step. */
int fdtd2d_step_observed(fdtd2d_handle h, int n0, int count, double weight, int window, void* stream);
int fdtd2d_observe_rms_window(fdtd2d_handle h, double* host, size_t n);
/* Waits for everything queued on `stream` (stepping calls return once their work is queued; downloads wait by themselves). */
int fdtd2d_sync(fdtd2d_handle h, void* stream);
size_t fdtd2d_device_bytes(fdtd2d_handle h);
/* Probe time series, as in 3-D: cells (i, k) of the field arrays (ids 0..20) or C1..C3; fdtd2d_step
 * records one sample per step. */
int fdtd2d_probe_add(fdtd2d_handle h, int id, int i, int k);
int fdtd2d_probe_record(fdtd2d_handle h, void* stream);
int fdtd2d_probe_read(fdtd2d_handle h, double* out, size_t cap, int* n_samples, int* n_probes);
int fdtd2d_probe_reset(fdtd2d_handle h, int drop_probes);
/* Kernel timing as in 3-D; kinds: 0 = E part, 1 = H part, 2 = fused step, 3 = two steps per launch. */
int fdtd2d_kernel_timing(fdtd2d_handle h, int enable);
int fdtd2d_kernel_time(fdtd2d_handle h, int kind, int* launches, double* total_ms);

#ifdef __cplusplus
}
#endif
#endif /* FDTD_ENGINE_H */
