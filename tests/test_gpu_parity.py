"""HIP engine against the CPU twin and the golden vectors, through the C ABI (needs a GPU)."""
import numpy as np
import pytest

import fdtd_amd
from oracle import pyoracle as po
from tests import util

pytestmark = pytest.mark.gpu

FIX3D = ["g1_3d_cpml", "g2_3d_cavity", "g3_3d_hetero"]


def engine3d_from_meta(meta, prec, **kw):
    return fdtd_amd.Engine3D(meta["N"], meta["npml"], (meta["dx"], meta["dy"], meta["dz"]), meta["dt"], meta["mu0"],
                             meta["k_Jx"], meta["steps"], precision=prec, **kw)


def engine2d_from_meta(meta, prec):
    return fdtd_amd.Engine2D(meta["Nx"], meta["Nz"], meta["npml"], meta["dx"], meta["dz"], meta["dt"], meta["mu0"],
                             meta["eps0"], meta["i_Jy"], meta["k_Jy"], meta["steps"], precision=prec)


@pytest.mark.parametrize("name", FIX3D)
def test_fp64_engine_equals_reference_golden_3d(name):
    """fp64 instantiation of the kernels == compiled reference, bit for bit (fields and CPML state)."""
    meta, ins, outs, extra = util.load_fixture(name)
    eng = engine3d_from_meta(meta, 64)
    util.feed(eng, ins)
    got = util.run_snapshots(eng, meta, outs)
    for key, want in outs.items():
        util.assert_equal_arrays(got[key], want, f"{name} step {key[0]} array {key[1]}")


@pytest.mark.parametrize("name", FIX3D)
def test_fp32_engine_equals_fp32_twin_3d(name):
    """fp32 kernels == fp32 twin, bit for bit, and within the stated drift of the fp64 reference."""
    meta, ins, outs, extra = util.load_fixture(name)
    eng = engine3d_from_meta(meta, 32)
    tw = util.twin3d_from_meta(meta, 32)
    util.feed(eng, ins)
    util.feed(tw, ins)
    got = util.run_snapshots(eng, meta, outs)
    ref = util.run_snapshots(tw, meta, outs)
    for key in outs:
        util.assert_equal_arrays(got[key], ref[key], f"{name} step {key[0]} array {key[1]}")
    for step in meta["snapshots"]:
        for fam in range(4):
            ids = [fam * 3 + c for c in range(3)]
            # stated tolerance vs the fp64 reference: 1e-6 (D,E) / 3e-6 (B,H) relative L2, <= 120 steps
            assert util.family_rel_l2([got[(step, i)] for i in ids], [outs[(step, i)] for i in ids]) < (1e-6 if fam < 2 else 3e-6)


@pytest.mark.parametrize("name", FIX3D)
@pytest.mark.parametrize("prec", [32, 64])
def test_lean_storage_equals_full_storage_3d(name, prec):
    """Lean storage (E = D/eps, H = B/mu recomputed where consumed) must give the same bits:
    fp64 against the reference vectors, fp32 against the fp32 twin."""
    meta, ins, outs, _ = util.load_fixture(name)
    eng = engine3d_from_meta(meta, prec, mode=fdtd_amd.engine.MODE_LEAN)
    util.feed(eng, ins)
    got = util.run_snapshots(eng, meta, outs)
    if prec == 64:
        ref = outs
    else:
        tw = util.twin3d_from_meta(meta, 32)
        util.feed(tw, ins)
        ref = util.run_snapshots(tw, meta, outs)
    for key in outs:
        util.assert_equal_arrays(got[key], ref[key], f"{name} lean fp{prec} step {key[0]} array {key[1]}")


def test_probe_series_matches_reference_3d():
    meta, ins, outs, extra = util.load_fixture("g2_3d_cavity")
    eng = engine3d_from_meta(meta, 64)
    util.feed(eng, ins)
    cell = tuple(meta["probe_cell"])
    shp = po.shape3d(po.EX, meta["N"], meta["npml"])
    probe = []
    for n in range(meta["steps"]):
        eng.step(n, 1)
        probe.append(eng.get(po.EX, shp)[cell])
    util.assert_equal_arrays(np.array(probe), extra["probe"], "probe series")


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("prec", [32, 64])
def test_mu_grid_3d(prec, mode):
    """Heterogeneous mu has no reference counterpart: checked against the twin, and mu == mu0 must
    reproduce the reference vectors."""
    meta, ins, outs, _ = util.load_fixture("g3_3d_hetero")
    eng = engine3d_from_meta(meta, prec, has_mu=True, mode=mode)
    tw = util.twin3d_from_meta(meta, prec, has_mu=True)
    util.feed(eng, ins)
    util.feed(tw, ins)
    if prec == 64:
        for c in range(3):
            mu = np.full(eng.size(po.MU_X + c), meta["mu0"])
            eng.set(po.MU_X + c, mu)
        got = util.run_snapshots(eng, meta, outs)
        for key, want in outs.items():
            util.assert_equal_arrays(got[key], want, f"mu==mu0 array {key[1]}")
        return
    eng.fill_materials(77, meta["eps0"], 4 * meta["eps0"], meta["mu0"], 2 * meta["mu0"])
    tw.fill_materials(77, meta["eps0"], 4 * meta["eps0"], meta["mu0"], 2 * meta["mu0"])
    for id_ in (po.EPS_X, po.EPS_Y, po.EPS_Z, po.MU_X, po.MU_Y, po.MU_Z):
        want = tw.get(id_, po.shape3d(id_, meta["N"], meta["npml"])).copy()
        if want.shape[0] == meta["N"][0] + 1:
            want[-1] = 0            # the wall column i = Nx is not stored by the engine (never read either)
        util.assert_equal_arrays(eng.get(id_), want, f"device-filled material {id_}")
    eng.step(0, meta["steps"])
    tw.step(0, meta["steps"])
    for id_ in po.FIELDS3D + po.ALL_PHI3D:
        util.assert_equal_arrays(eng.get(id_), tw.get(id_), f"mu grid array {id_}")


@pytest.mark.parametrize("N,npml", [((300, 9, 11), (5, 6, 2, 2, 3, 3)), ((7, 6, 5), (0, 0, 0, 0, 0, 0)),
                                    ((258, 5, 6), (0, 3, 1, 0, 0, 2)), ((12, 10, 9), (4, 4, 4, 4, 4, 4))])
def test_ragged_shapes_3d(N, npml):
    """Grids that are not multiples of the tile, more than one x tile, tiny grids, one-sided CPML."""
    K = po.consts()
    d = (5e-8, 6e-8, 7e-8)
    dt = 5e-8 / (2 * K["c0"])
    steps = 25
    kJ = N[2] // 2
    eng = fdtd_amd.Engine3D(N, npml, d, dt, K["mu0"], kJ, steps, precision=32)
    tw = po.Twin3D(32, N, npml, d, dt, K["mu0"], kJ, steps)
    rng = np.random.default_rng(5)
    for be in (eng, tw):
        for a in range(3):
            for s in range(2):
                tabs = po.cpml_face(npml[a * 2 + s], s, d[a], d[a], dt)
                for k in range(6):
                    if npml[a * 2 + s]:
                        be.set(po.tab_id(a, s, k), tabs[k])
    wf, mag = po.default_source3d(N[0], N[1], steps, d, dt)
    mag = rng.random(mag.shape)
    eps = [K["eps0"] * (1 + 3 * rng.random(po.shape3d(po.EPS_X + c, N, npml))) for c in range(3)]
    for be in (eng, tw):
        be.set(po.JX_WF, wf)
        be.set(po.JX_MAG, mag)
        for c in range(3):
            be.set(po.EPS_X + c, eps[c])
    eng.step(0, steps)
    tw.step(0, steps)
    for id_ in po.FIELDS3D + po.ALL_PHI3D:
        if eng.size(id_):
            util.assert_equal_arrays(eng.get(id_), tw.get(id_), f"N={N} array {id_}")


def test_set_get_roundtrip_3d():
    N, npml = (20, 7, 9), (3, 2, 2, 3, 1, 4)
    K = po.consts()
    eng = fdtd_amd.Engine3D(N, npml, (1e-7,) * 3, 1e-16, K["mu0"], 4, 3, precision=64)
    rng = np.random.default_rng(1)
    for id_ in po.FIELDS3D + [po.EPS_X, po.EPS_Y, po.EPS_Z, po.JX_MAG]:
        a = rng.standard_normal(po.shape3d(id_, N, npml))
        if id_ < 18 and id_ % 3 != 0 and (id_ // 3) in (0, 1, 4):
            a[-1, :, :] = 0          # wall column i = Nx of the y/z electric components is not stored
        if id_ in (po.BX, po.HX):
            a[-1, :, :] = 0
        eng.set(id_, a)
        util.assert_equal_arrays(eng.get(id_), a, f"roundtrip {id_}")
    for id_ in po.ALL_PHI3D:
        assert eng.size(id_) == int(np.prod(po.shape3d(id_, N, npml)))


def test_fp64_engine_equals_reference_golden_2d():
    meta, ins, outs, _ = util.load_fixture("g4_2d_slab")
    eng = engine2d_from_meta(meta, 64)
    util.feed(eng, ins)
    got = util.run_snapshots(eng, meta, outs)
    for key, want in outs.items():
        util.assert_equal_arrays(got[key], want, f"g4 array {key[1]}")


def test_fp32_engine_equals_fp32_twin_2d():
    meta, ins, outs, _ = util.load_fixture("g4_2d_slab")
    eng = engine2d_from_meta(meta, 32)
    tw = util.twin2d_from_meta(meta, 32)
    util.feed(eng, ins)
    util.feed(tw, ins)
    got = util.run_snapshots(eng, meta, outs)
    ref = util.run_snapshots(tw, meta, outs)
    for key in outs:
        util.assert_equal_arrays(got[key], ref[key], f"g4 fp32 array {key[1]}")
    for (step, id_), want in outs.items():
        if id_ in (po.D2_DY, po.D2_EY, po.D2_BX, po.D2_BZ, po.D2_HX, po.D2_HZ):
            assert util.rel_l2(got[(step, id_)], want) < 5e-6     # stated tolerance vs fp64 reference, 160 steps


def test_pec512_point_source_2d():
    """The 512x512 PEC point-source case on the GPU: snapshots, probe series and sums == reference (fp64)."""
    meta, ins, outs, extra = util.load_fixture("g5_2d_pec512")
    eng = engine2d_from_meta(meta, 64)
    util.feed(eng, ins)
    shp = po.shape2d(po.D2_EY, meta["Nx"], meta["Nz"], meta["npml"])
    sub = meta["subsample"]
    n = 0
    for snap in meta["snapshots"]:
        eng.step(n, snap - n)
        n = snap
        ey = eng.get(po.D2_EY, shp)
        util.assert_equal_arrays(ey[::sub, ::sub], outs[(snap, po.D2_EY)], f"g5 Ey at {snap}")
        assert util.seq_sumsq(ey) == meta["sumsq_ey"][str(snap)]
        assert abs(eng.sumsq(po.D2_EY) / meta["sumsq_ey"][str(snap)] - 1) < 1e-12


@pytest.mark.parametrize("Nx,Nz,npml", [(300, 37, (4, 5, 3, 2)), (50, 45, (0, 0, 0, 0)), (1030, 20, (6, 0, 0, 5))])
def test_ragged_shapes_2d(Nx, Nz, npml):
    K = po.consts()
    dx, dz = 5e-8, 6e-8
    dt = dx / (2 * K["c0"])
    steps = 60
    eng = fdtd_amd.Engine2D(Nx, Nz, npml, dx, dz, dt, K["mu0"], K["eps0"], Nx // 3, Nz // 2, steps, precision=32)
    tw = po.Twin2D(32, Nx, Nz, npml, dx, dz, dt, K["mu0"], K["eps0"], Nx // 3, Nz // 2, steps)
    rng = np.random.default_rng(9)
    C = [0.02 * rng.random((Nx, Nz)), -1.9 + 0.01 * rng.random((Nx, Nz)), 1 + 0.01 * rng.random((Nx, Nz))]
    jy = po.default_source2d(steps, dx, dz, dt)
    dd = (dx, dz)
    for be in (eng, tw):
        for a in range(2):
            for s in range(2):
                if npml[a * 2 + s]:
                    tabs = po.cpml_face(npml[a * 2 + s], s, dd[a], dz, dt)
                    for k in range(6):
                        be.set(po.D2_TAB + (a * 2 + s) * 6 + k, tabs[k])
        for q in range(3):
            be.set(po.D2_C1 + q, C[q])
        be.set(po.D2_JY, jy)
    eng.step(0, steps)
    tw.step(0, steps)
    for id_ in po.FIELDS2D + po.ALL_PHI2D:
        if eng.size(id_):
            util.assert_equal_arrays(eng.get(id_), tw.get(id_), f"2-D {Nx}x{Nz} array {id_}")


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_device_probe_series_3d(mode):
    """Probe time series recorded on the device == the reference's series at the same cell (fp64), plus
    a B-family and a D-family probe against a per-step download; every storage mode."""
    meta, ins, outs, extra = util.load_fixture("g2_3d_cavity")
    eng = engine3d_from_meta(meta, 64, mode=mode)
    util.feed(eng, ins)
    cell = tuple(meta["probe_cell"])
    eng.probe_add(po.EX, *cell)
    eng.probe_add(po.HY, 3, 4, 5)
    eng.probe_add(po.DZ, 5, 6, 7)
    with pytest.raises(fdtd_amd.engine.EngineError):
        eng.probe_add(po.EY, meta["N"][0], 0, 0)          # wall column
    with pytest.raises(fdtd_amd.engine.EngineError):
        eng.probe_add(po.EX, 0, 0, meta["N"][2] + 5)       # outside
    eng.step(0, 50)
    hy50 = eng.get(po.HY, po.shape3d(po.HY, meta["N"], meta["npml"]))[3, 4, 5]
    eng.step(50, meta["steps"] - 50)
    series = eng.probe_read()
    assert series.shape == (meta["steps"], 3)
    util.assert_equal_arrays(series[:, 0], extra["probe"], "device probe series")
    assert series[49, 1] == hy50
    assert series[-1, 2] == eng.get(po.DZ, po.shape3d(po.DZ, meta["N"], meta["npml"]))[5, 6, 7]
    with pytest.raises(fdtd_amd.engine.EngineError):
        eng.probe_add(po.EX, 1, 1, 1)                      # samples exist: needs a reset first
    eng.probe_reset(False)
    assert eng.probe_read().shape == (0, 3)


def test_device_probe_series_2d():
    meta, ins, outs, extra = util.load_fixture("g5_2d_pec512")
    eng = engine2d_from_meta(meta, 64)
    util.feed(eng, ins)
    eng.probe_add(po.D2_EY, *meta["probe_cell"])
    eng.step(0, meta["steps"])
    series = eng.probe_read()
    assert series.shape == (meta["steps"], 1)
    util.assert_equal_arrays(series[:, 0], extra["probe"], "2-D device probe series")


@pytest.mark.parametrize("z_range", [None, (30, 71)])
def test_reference_materials_on_device_equal_host_generator(z_range):
    """FDTD3D::setupMaterials evaluated on the device == the host restatement (itself pinned to the
    compiled reference in tests/test_oracle_ref.py), bit for bit, sphere surface included."""
    from fdtd_amd import setup as fs
    N = (96, 90, 100)
    d = (1053e-9 / 20,) * 3
    K = po.consts()
    eng = fdtd_amd.Engine3D(N, (0,) * 6, d, d[0] / (2 * K["c0"]), K["mu0"], 20, 1, precision=64, z_range=z_range)
    eng.fill_reference_materials()
    want = fs.default_materials3d(N, d)
    assert len({float(v) for v in np.unique(want[0])}) > 2          # the blend values of the sphere surface are present
    for c in range(3):
        w = want[c].copy()
        if w.shape[0] == N[0] + 1:
            w[-1] = 0                                               # unstored wall column
        got = np.full(w.size, np.nan)
        eng.get(po.EPS_X + c, out=got)
        got = got.reshape(w.shape)
        if z_range is not None:
            lo, hi = z_range
            util.assert_equal_arrays(got[:, :, lo:hi], w[:, :, lo:hi], f"device eps component {c}, slab")
        else:
            util.assert_equal_arrays(got, w, f"device eps component {c}")


@pytest.mark.parametrize("seed", range(12))
def test_random_configurations_2d_equal_twin(seed):
    """Random 2-D shapes (one to three x tiles), per-side CPML counts, source cell, precision, rows per
    wave, and Lorentz coefficients that are non-zero only in a random band of columns or rows -- so that
    dispersive and non-dispersive tiles meet inside the grid: every field, history and CPML array must
    equal the twin's bit for bit."""
    rng = np.random.default_rng(500 + seed)
    Nx = int(rng.choice([rng.integers(8, 60), rng.integers(250, 270), rng.integers(500, 700)]))
    Nz = int(rng.integers(6, 50))
    npml = [int(rng.integers(0, 6)) if rng.random() < 0.7 else 0 for _ in range(4)]
    while npml[0] + npml[1] >= max(Nx - 9, 1):
        npml[0], npml[1] = npml[0] // 2, npml[1] // 2
    while npml[2] + npml[3] >= Nz - 1:
        npml[2], npml[3] = npml[2] // 2, npml[3] // 2
    prec = int(rng.choice([32, 64]))
    K = po.consts()
    dx, dz = 5e-8, 6e-8
    dt = dx / (2 * K["c0"])
    steps = 40
    iJ, kJ = int(rng.integers(0, Nx)), int(rng.integers(0, Nz + 1))
    import os
    os.environ["FDTD_Z_CHUNK2D"] = str(int(rng.choice([1, 2, 5, 32])))
    try:
        eng = fdtd_amd.Engine2D(Nx, Nz, npml, dx, dz, dt, K["mu0"], K["eps0"], iJ, kJ, steps, precision=prec)
    finally:
        del os.environ["FDTD_Z_CHUNK2D"]
    tw = po.Twin2D(prec, Nx, Nz, npml, dx, dz, dt, K["mu0"], K["eps0"], iJ, kJ, steps)
    mask = np.zeros((Nx, Nz), dtype=bool)
    if rng.random() < 0.5:
        a = int(rng.integers(0, Nx)); mask[a:a + int(rng.integers(1, 40)), :] = True
    else:
        a = int(rng.integers(0, Nz)); mask[:, a:a + int(rng.integers(1, 6))] = True
    if seed % 4 == 3:
        mask[:] = False                                            # no dispersive cell at all
    Cs = [np.where(mask, 0.02 * rng.random((Nx, Nz)), 0.0), np.where(mask, -1.9 + 0.01 * rng.random((Nx, Nz)), 0.0),
          np.where(mask, 1 + 0.01 * rng.random((Nx, Nz)), 0.0)]
    jy = rng.standard_normal(steps) * 1e3
    dd = (dx, dz)
    for be in (eng, tw):
        for a in range(2):
            for s in range(2):
                if npml[a * 2 + s]:
                    tabs = po.cpml_face(npml[a * 2 + s], s, dd[a], dz, dt)
                    for k in range(6):
                        be.set(po.D2_TAB + (a * 2 + s) * 6 + k, tabs[k])
        for q in range(3):
            be.set(po.D2_C1 + q, Cs[q])
        be.set(po.D2_JY, jy)
    eng.step(0, steps)
    tw.step(0, steps)
    for id_ in po.FIELDS2D + po.ALL_PHI2D:
        if eng.size(id_):
            util.assert_equal_arrays(eng.get(id_), tw.get(id_), f"seed {seed}: 2-D {Nx}x{Nz} npml={npml} fp{prec} source=({iJ},{kJ}), array {id_}")


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("seed", range(8))
def test_random_configurations_half_step_kernels_equal_twin(seed, mode):
    """The two-kernel paths on the same kind of random problems the fused kernel is checked on."""
    from tests.test_gpu_fused import _random_case
    rng = np.random.default_rng(300 + seed)
    N, npml, prec, has_mu, rows, chunk = _random_case(rng)
    K = po.consts()
    d = (5e-8, 6e-8, 7e-8)
    dt = 5e-8 / (2 * K["c0"])
    steps = 13
    kJ = int(rng.integers(0, N[2] + 1))
    eng = fdtd_amd.Engine3D(N, npml, d, dt, K["mu0"], kJ, steps, precision=prec, mode=mode, has_mu=has_mu)
    eng.tune(chunk, int(rng.integers(0, 2)), 1 + rows % 4)
    tw = po.Twin3D(prec, N, npml, d, dt, K["mu0"], kJ, steps, has_mu)
    wf = rng.standard_normal(steps)
    mag = rng.random((N[0], N[1] + 1))
    for be in (eng, tw):
        for a in range(3):
            for s in range(2):
                if npml[a * 2 + s]:
                    tabs = po.cpml_face(npml[a * 2 + s], s, d[a], d[a], dt)
                    for k in range(6):
                        be.set(po.tab_id(a, s, k), tabs[k])
        be.set(po.JX_WF, wf)
        be.set(po.JX_MAG, mag)
        be.fill_materials(seed, K["eps0"], 4 * K["eps0"], K["mu0"], 3 * K["mu0"])
    eng.step(0, steps)
    tw.step(0, steps)
    for id_ in po.FIELDS3D + po.ALL_PHI3D:
        if eng.size(id_):
            util.assert_equal_arrays(eng.get(id_), tw.get(id_), f"seed {seed} mode {mode}: N={N} npml={npml} fp{prec} mu={has_mu}, array {id_}")


def test_tiny_grids_equal_twin():
    """Grids of 2, 3 and 5 cells per axis (the smallest the engine accepts), with and without a one-layer
    CPML, every storage mode, both precisions, with and without a mu grid: 648 problems == the twin."""
    import itertools
    K = po.consts()
    d = (5e-8, 6e-8, 7e-8)
    dt = 5e-8 / (2 * K["c0"])
    rng = np.random.default_rng(5)
    steps, cases = 6, 0
    for N in itertools.product((2, 3, 5), repeat=3):
        for layers in (0, 1):
            npml = tuple(layers if N[a // 2] > 2 * layers + 1 else 0 for a in range(6))
            for mode, prec, has_mu in itertools.product((0, 1, 2), (32, 64), (False, True)):
                kJ = int(rng.integers(0, N[2] + 1))
                eng = fdtd_amd.Engine3D(N, npml, d, dt, K["mu0"], kJ, steps, precision=prec, mode=mode, has_mu=has_mu)
                tw = po.Twin3D(prec, N, npml, d, dt, K["mu0"], kJ, steps, has_mu)
                wf, mag = rng.standard_normal(steps), rng.random((N[0], N[1] + 1))
                for be in (eng, tw):
                    for a in range(3):
                        for s_ in range(2):
                            if npml[a * 2 + s_]:
                                tabs = po.cpml_face(npml[a * 2 + s_], s_, d[a], d[a], dt)
                                for k in range(6):
                                    be.set(po.tab_id(a, s_, k), tabs[k])
                    be.set(po.JX_WF, wf)
                    be.set(po.JX_MAG, mag)
                    be.fill_materials(7, K["eps0"], 4 * K["eps0"], K["mu0"], 3 * K["mu0"])
                eng.step(0, steps)
                tw.step(0, steps)
                for id_ in po.FIELDS3D + po.ALL_PHI3D:
                    if eng.size(id_):
                        util.assert_equal_arrays(eng.get(id_), tw.get(id_), f"N={N} npml={npml} mode={mode} fp{prec} mu={has_mu}, array {id_}")
                eng.close()
                cases += 1
    assert cases == 648


@pytest.mark.parametrize("N,npml", [((140, 40, 150), (64, 60, 10, 12, 64, 64)), ((40, 140, 70), (5, 0, 64, 64, 33, 30)),
                                    ((300, 20, 66), (64, 64, 3, 3, 64, 1)), ((64, 64, 64), (31, 32, 31, 32, 31, 32))])
def test_maximum_cpml_layers_equal_twin(N, npml):
    """Up to 64 layers per face (the table limit), CPML filling the grid but for one cell, strips wider
    than an x tile: every storage mode and both precisions == the twin."""
    import itertools
    K = po.consts()
    d = (5e-8, 6e-8, 7e-8)
    dt = 5e-8 / (2 * K["c0"])
    rng = np.random.default_rng(3)
    steps = 7
    for mode, prec in itertools.product((0, 1, 2), (32, 64)):
        kJ, has_mu = int(rng.integers(0, N[2] + 1)), bool(rng.integers(0, 2))
        eng = fdtd_amd.Engine3D(N, npml, d, dt, K["mu0"], kJ, steps, precision=prec, mode=mode, has_mu=has_mu)
        tw = po.Twin3D(prec, N, npml, d, dt, K["mu0"], kJ, steps, has_mu)
        wf, mag = rng.standard_normal(steps), rng.random((N[0], N[1] + 1))
        for be in (eng, tw):
            for a in range(3):
                for s_ in range(2):
                    if npml[a * 2 + s_]:
                        tabs = po.cpml_face(npml[a * 2 + s_], s_, d[a], d[a], dt)
                        for k in range(6):
                            be.set(po.tab_id(a, s_, k), tabs[k])
            be.set(po.JX_WF, wf)
            be.set(po.JX_MAG, mag)
            be.fill_materials(7, K["eps0"], 4 * K["eps0"], K["mu0"], 3 * K["mu0"])
        eng.step(0, steps)
        tw.step(0, steps)
        for id_ in po.FIELDS3D + po.ALL_PHI3D:
            if eng.size(id_):
                util.assert_equal_arrays(eng.get(id_), tw.get(id_), f"N={N} npml={npml} mode={mode} fp{prec} mu={has_mu}, array {id_}")
        eng.close()
