"""The one-kernel 2-D step (FDTD2D_MODE_FUSED) against the golden vectors and the twin (needs a GPU).

In this mode Ey, Hx, Hz only live in memory at cells with a non-zero Lorentz coefficient; downloads of
the current fields derive the rest from Dy/eps0 and B/mu0.  History samples and polarisation sums are
therefore compared at the dispersive cells only -- the recursion never reads them anywhere else."""
import os

import numpy as np
import pytest

import fdtd_amd
from fdtd_amd.engine import MODE2D_FUSED
from oracle import pyoracle as po
from tests import util

pytestmark = pytest.mark.gpu

CURRENT = (po.D2_DY, po.D2_EY, po.D2_BX, po.D2_HX, po.D2_BZ, po.D2_HZ)


@pytest.fixture(autouse=True, params=[None, 1, 3, 32], ids=["one-step", "two-step-chunk1", "two-step-chunk3", "two-step-chunk32"])
def two_steps_per_launch(request, monkeypatch):
    """Every test of this file also runs with fdtd2d_step taking its steps two per launch (k2_pair; grids this small
    only do so when FDTD2D_PAIR=1 says so), in chunks of 1, 3 and 32 rows per wave: each must leave what single steps
    leave, bit for bit -- fields, samples, polarisation sums and the CPML state of both sides."""
    if request.param is not None:
        monkeypatch.setenv("FDTD2D_PAIR", "1")
        monkeypatch.setenv("FDTD_Z_CHUNK2D_PAIR", str(request.param))
    return request.param


def compare_2d(eng, ref_get, Nx, Nz, npml, mask, what):
    """mask: [Nx][Nz] bool, cells with a non-zero coefficient."""
    for id_ in po.FIELDS2D + po.ALL_PHI2D:
        if not eng.size(id_):
            continue
        shp = po.shape2d(id_, Nx, Nz, npml)
        got, want = eng.get(id_, shp), np.asarray(ref_get(id_)).reshape(shp)
        if id_ in CURRENT or id_ >= 100 or id_ % 7 == 2:          # current fields (n1 is the current sample), CPML state
            util.assert_equal_arrays(got, want, f"{what} array {id_}")
        else:                                                      # older samples and polarisation sums: dispersive cells only
            g, w = got[:Nx, :Nz][mask], want[:Nx, :Nz][mask]
            util.assert_equal_arrays(g, w, f"{what} array {id_} at dispersive cells")


@pytest.mark.parametrize("prec", [32, 64])
@pytest.mark.parametrize("chunk", [1, 7, 64])
def test_fused2d_lorentz_slab(prec, chunk):
    """The reference's default slab problem (golden g4): fp64 == compiled reference, fp32 == fp32 twin."""
    meta, ins, outs, _ = util.load_fixture("g4_2d_slab")
    os.environ["FDTD_Z_CHUNK2D"] = str(chunk)
    try:
        eng = fdtd_amd.Engine2D(meta["Nx"], meta["Nz"], meta["npml"], meta["dx"], meta["dz"], meta["dt"], meta["mu0"], meta["eps0"],
                                meta["i_Jy"], meta["k_Jy"], meta["steps"], precision=prec, mode=MODE2D_FUSED)
    finally:
        del os.environ["FDTD_Z_CHUNK2D"]
    util.feed(eng, ins)
    eng.step(0, meta["steps"])
    mask = np.asarray(ins[po.D2_C1]).reshape(meta["Nx"], meta["Nz"]) != 0
    assert mask.any() and not mask.all()
    if prec == 64:
        ref_get = lambda id_: outs[(meta["steps"], id_)]
    else:
        tw = util.twin2d_from_meta(meta, 32)
        util.feed(tw, ins)
        tw.step(0, meta["steps"])
        ref_get = tw.get
    compare_2d(eng, ref_get, meta["Nx"], meta["Nz"], meta["npml"], mask, f"g4 fused fp{prec} chunk {chunk}")


def test_fused2d_pec512_point_source():
    """512 x 512, PEC walls, no dispersive cell, 2000 steps: snapshots, sums and the probe series == reference (fp64)."""
    meta, ins, outs, extra = util.load_fixture("g5_2d_pec512")
    eng = fdtd_amd.Engine2D(meta["Nx"], meta["Nz"], meta["npml"], meta["dx"], meta["dz"], meta["dt"], meta["mu0"], meta["eps0"],
                            meta["i_Jy"], meta["k_Jy"], meta["steps"], precision=64, mode=MODE2D_FUSED)
    util.feed(eng, ins)
    eng.probe_add(po.D2_EY, *meta["probe_cell"])
    shp = po.shape2d(po.D2_EY, meta["Nx"], meta["Nz"], meta["npml"])
    sub = meta["subsample"]
    n = 0
    for snap in meta["snapshots"]:
        eng.step(n, snap - n)
        n = snap
        ey = eng.get(po.D2_EY, shp)
        util.assert_equal_arrays(ey[::sub, ::sub], outs[(snap, po.D2_EY)], f"g5 fused Ey at {snap}")
        assert util.seq_sumsq(ey) == meta["sumsq_ey"][str(snap)]
        assert abs(eng.sumsq(po.D2_EY) / meta["sumsq_ey"][str(snap)] - 1) < 1e-12
    util.assert_equal_arrays(eng.probe_read()[:, 0], extra["probe"], "g5 fused probe series")


@pytest.mark.parametrize("seed", range(16))
def test_fused2d_random_configurations_equal_twin(seed):
    rng = np.random.default_rng(900 + seed)
    Nx = int(rng.choice([rng.integers(8, 60), rng.integers(245, 262), rng.integers(500, 700)]))
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
    steps = 41
    iJ, kJ = int(rng.integers(0, Nx)), int(rng.integers(0, Nz + 1))
    os.environ["FDTD_Z_CHUNK2D"] = str(int(rng.choice([1, 2, 5, 16, 100])))
    try:
        eng = fdtd_amd.Engine2D(Nx, Nz, npml, dx, dz, dt, K["mu0"], K["eps0"], iJ, kJ, steps, precision=prec, mode=MODE2D_FUSED)
    finally:
        del os.environ["FDTD_Z_CHUNK2D"]
    tw = po.Twin2D(prec, Nx, Nz, npml, dx, dz, dt, K["mu0"], K["eps0"], iJ, kJ, steps)
    mask = np.zeros((Nx, Nz), dtype=bool)
    kind = seed % 4
    if kind == 0:
        a = int(rng.integers(0, Nx)); mask[a:a + int(rng.integers(1, 40)), :] = True
    elif kind == 1:
        a = int(rng.integers(0, Nz)); mask[:, a:a + int(rng.integers(1, 6))] = True
    elif kind == 2:
        mask[rng.random((Nx, Nz)) < 0.02] = True                   # isolated dispersive cells
    Cs = [np.where(mask, 0.02 * rng.random((Nx, Nz)), 0.0), np.where(mask, -1.9 + 0.01 * rng.random((Nx, Nz)), 0.0),
          np.where(mask, 1 + 0.01 * rng.random((Nx, Nz)), 0.0)]
    mask = (Cs[0] != 0) | (Cs[1] != 0) | (Cs[2] != 0)
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
    compare_2d(eng, tw.get, Nx, Nz, npml, mask, f"seed {seed}: 2-D fused {Nx}x{Nz} npml={npml} fp{prec} source=({iJ},{kJ})")


def test_two_step_launches_are_what_runs(two_steps_per_launch):
    """7 steps = three launches of the two-step kernel and one of the one-step kernel when pairing is on, seven one-step
    launches otherwise; with a probe registered every step is a launch of its own (a probe is sampled after every step)."""
    meta, ins, _, _ = util.load_fixture("g4_2d_slab")
    make = lambda: fdtd_amd.Engine2D(meta["Nx"], meta["Nz"], meta["npml"], meta["dx"], meta["dz"], meta["dt"], meta["mu0"], meta["eps0"],
                                     meta["i_Jy"], meta["k_Jy"], meta["steps"], precision=32, mode=MODE2D_FUSED)
    eng = make()
    util.feed(eng, ins)
    eng.kernel_timing(True)
    eng.step(0, 7)
    eng.kernel_timing(False)
    pairs, singles = eng.kernel_time(3)[0], eng.kernel_time(2)[0]
    assert (pairs, singles) == ((3, 1) if two_steps_per_launch is not None else (0, 7))
    probed = make()
    util.feed(probed, ins)
    probed.probe_add(po.D2_DY, 3, 4)
    probed.kernel_timing(True)
    probed.step(0, 7)
    probed.kernel_timing(False)
    assert (probed.kernel_time(3)[0], probed.kernel_time(2)[0]) == (0, 7)
    util.assert_equal_arrays(probed.get(po.D2_DY), eng.get(po.D2_DY), "Dy after 7 steps, with and without a probe")
    assert probed.probe_read().shape[0] == 7


def test_fused2d_observables_equal_twin():
    meta, ins, _, _ = util.load_fixture("g4_2d_slab")
    eng = fdtd_amd.Engine2D(meta["Nx"], meta["Nz"], meta["npml"], meta["dx"], meta["dz"], meta["dt"], meta["mu0"], meta["eps0"],
                            meta["i_Jy"], meta["k_Jy"], meta["steps"], precision=32, mode=MODE2D_FUSED)
    tw = util.twin2d_from_meta(meta, 32)
    util.feed(eng, ins)
    util.feed(tw, ins)
    for n in range(30):
        eng.step(n, 1)
        tw.step(n, 1)
        eng.observe_accumulate(0.05, n % 10 == 0)
        tw.observe_accumulate(0.05, n % 10 == 0)
    util.assert_equal_arrays(eng.observe_rms(), tw.observe_rms(), "2-D fused RMS window")


@pytest.mark.parametrize("prec", [32, 64])
@pytest.mark.parametrize("window", [1, 2, 5])
def test_window_inside_two_step_launches_2d(prec, window):
    """fdtd2d_step_observed on the reference's slab problem: calls of odd and even length starting at odd and even steps;
    the window being accumulated and the last completed one equal stepping and accumulating one step at a time on the
    twin, bit for bit (with two steps per launch the window is advanced by the step kernel itself)."""
    meta, ins, _, _ = util.load_fixture("g4_2d_slab")
    eng = fdtd_amd.Engine2D(meta["Nx"], meta["Nz"], meta["npml"], meta["dx"], meta["dz"], meta["dt"], meta["mu0"], meta["eps0"],
                            meta["i_Jy"], meta["k_Jy"], meta["steps"], precision=prec, mode=MODE2D_FUSED)
    tw = util.twin2d_from_meta(meta, prec)
    util.feed(eng, ins)
    util.feed(tw, ins)
    with pytest.raises(fdtd_amd.engine.EngineError, match="no window has been completed"):
        eng.observe_rms_window()
    n, kept = 0, None
    for count in (1, 6, 7, 2, 9):
        eng.step_observed(n, count, 0.05, window)
        for m in range(n, n + count):
            tw.step(m, 1)
            tw.observe_accumulate(0.05, m % window == 0)
            if (m + 1) % window == 0:
                kept = tw.observe_rms().copy()
        n += count
        util.assert_equal_arrays(eng.observe_rms(), tw.observe_rms(), f"fp{prec} window {window}: accumulating, after step {n - 1}")
        if kept is not None:
            util.assert_equal_arrays(eng.observe_rms_window(), kept, f"fp{prec} window {window}: completed, after step {n - 1}")
    assert kept is not None and kept.max() > 0
    mask = np.asarray(ins[po.D2_C1]).reshape(meta["Nx"], meta["Nz"]) != 0
    compare_2d(eng, tw.get, meta["Nx"], meta["Nz"], meta["npml"], mask, f"fp{prec} after {n} observed steps")


def test_short_division_by_a_constant_is_proven_and_the_proof_can_fail(monkeypatch):
    """The two-step kernel divides by eps0 and mu0 with x*r corrected by two fused multiply-adds where the library has
    proven, over every fp32 numerator of the class it uses it for, that this is the IEEE quotient.  The proof holds for the
    constants of the reference and for a handful of others; run over ALL magnitudes instead (denormals, values whose
    quotient overflows) it finds disagreements, i.e. it is a test that can fail; and a run with the short form switched off
    leaves the same bits."""
    import ctypes
    L = fdtd_amd.lib()
    L.fdtd_const_division_proven.argtypes = [ctypes.c_double]
    L.fdtd_const_division_proven.restype = ctypes.c_int
    L.fdtd_const_division_mismatches.argtypes = [ctypes.c_double] * 3
    L.fdtd_const_division_mismatches.restype = ctypes.c_longlong
    K = po.consts()
    for c in (K["eps0"], K["mu0"], 1.0, 3.0, 0.1, 7.77e5, 1.2345e-3):
        assert L.fdtd_const_division_proven(c) == 1, c
        lo, hi = 2.0 ** -80 * max(1.0, c), 2.0 ** 80 * min(1.0, c)
        assert L.fdtd_const_division_mismatches(c, lo, hi) == 0
    fmax = float(np.finfo(np.float32).max)
    assert L.fdtd_const_division_mismatches(K["eps0"], 0.0, fmax) > 0          # quotients overflow, tiny remainders are inexact
    assert L.fdtd_const_division_mismatches(3.0, 0.0, 2.0 ** 80) > 0           # denormal numerators
    meta, ins, _, _ = util.load_fixture("g4_2d_slab")
    got = []
    for switch in ("1", "0"):
        monkeypatch.setenv("FDTD_CONST_DIV", switch)
        monkeypatch.setenv("FDTD2D_PAIR", "1")
        eng = fdtd_amd.Engine2D(meta["Nx"], meta["Nz"], meta["npml"], meta["dx"], meta["dz"], meta["dt"], meta["mu0"], meta["eps0"],
                                meta["i_Jy"], meta["k_Jy"], meta["steps"], precision=32, mode=MODE2D_FUSED)
        util.feed(eng, ins)
        eng.step(0, 40)
        got.append([eng.get(i) for i in (po.D2_DY, po.D2_BX, po.D2_BZ)])
    for a_, b_ in zip(*got):
        assert np.array_equal(a_.view(np.uint64), b_.view(np.uint64))


def test_observed_steps_two_kernel_mode_2d():
    """Engines that keep every array of the reference (two kernels per step) take the same call: step, then the observe
    kernel, and the completed window is kept the same way."""
    meta, ins, _, _ = util.load_fixture("g4_2d_slab")
    eng = fdtd_amd.Engine2D(meta["Nx"], meta["Nz"], meta["npml"], meta["dx"], meta["dz"], meta["dt"], meta["mu0"], meta["eps0"],
                            meta["i_Jy"], meta["k_Jy"], meta["steps"], precision=64, mode=0)
    tw = util.twin2d_from_meta(meta, 64)
    util.feed(eng, ins)
    util.feed(tw, ins)
    eng.step_observed(0, 11, 0.1, 4)
    kept = None
    for n in range(11):
        tw.step(n, 1)
        tw.observe_accumulate(0.1, n % 4 == 0)
        if (n + 1) % 4 == 0:
            kept = tw.observe_rms().copy()
    util.assert_equal_arrays(eng.observe_rms(), tw.observe_rms(), "window being accumulated")
    util.assert_equal_arrays(eng.observe_rms_window(), kept, "last completed window")
    with pytest.raises(fdtd_amd.engine.EngineError, match="at least one step"):
        eng.step_observed(11, 2, 0.1, 0)


@pytest.mark.parametrize("seed", range(8))
def test_window_random_configurations_2d(seed):
    """Random 2-D grids (one to several x tiles), CPML layers, dispersive cells, window and run lengths."""
    rng = np.random.default_rng(7000 + seed)
    Nx = int(rng.choice([rng.integers(8, 60), rng.integers(120, 140), rng.integers(245, 262), rng.integers(300, 500)]))
    Nz = int(rng.integers(6, 50))
    npml = [int(rng.integers(0, 6)) if rng.random() < 0.7 else 0 for _ in range(4)]
    while npml[0] + npml[1] >= max(Nx - 9, 1):
        npml[0], npml[1] = npml[0] // 2, npml[1] // 2
    while npml[2] + npml[3] >= Nz - 1:
        npml[2], npml[3] = npml[2] // 2, npml[3] // 2
    prec = int(rng.choice([32, 64]))
    window = int(rng.choice([1, 2, 3, 4, 7]))
    counts = [int(rng.integers(1, 9)) for _ in range(3)]
    steps = sum(counts)
    K = po.consts()
    dx, dz = 5e-8, 6e-8
    dt = dx / (2 * K["c0"])
    iJ, kJ = int(rng.integers(0, Nx)), int(rng.integers(0, Nz + 1))
    eng = fdtd_amd.Engine2D(Nx, Nz, npml, dx, dz, dt, K["mu0"], K["eps0"], iJ, kJ, steps, precision=prec, mode=MODE2D_FUSED)
    tw = po.Twin2D(prec, Nx, Nz, npml, dx, dz, dt, K["mu0"], K["eps0"], iJ, kJ, steps)
    mask = rng.random((Nx, Nz)) < (0.02 if seed % 2 else 0.0)
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
    what = f"seed {seed}: {Nx}x{Nz} npml={npml} fp{prec} window {window} counts {counts}"
    n, kept = 0, None
    for count in counts:
        eng.step_observed(n, count, 0.3, window)
        for m in range(n, n + count):
            tw.step(m, 1)
            tw.observe_accumulate(0.3, m % window == 0)
            if (m + 1) % window == 0:
                kept = tw.observe_rms().copy()
        n += count
        util.assert_equal_arrays(eng.observe_rms(), tw.observe_rms(), f"{what}: accumulating after step {n - 1}")
        if kept is not None:
            util.assert_equal_arrays(eng.observe_rms_window(), kept, f"{what}: completed window after step {n - 1}")
    compare_2d(eng, tw.get, Nx, Nz, npml, mask, what)


def test_tiny_grids_2d_equal_twin():
    """2-D grids of 2, 3, 5 and 9 cells per axis, with and without a one-layer CPML, both step paths, both
    precisions, with and without scattered dispersive cells: 256 problems == the twin."""
    import itertools
    K = po.consts()
    dx, dz = 5e-8, 6e-8
    dt = dx / (2 * K["c0"])
    rng = np.random.default_rng(11)
    steps, cases = 12, 0
    for Nx, Nz in itertools.product((2, 3, 5, 9), repeat=2):
        for layers in (0, 1):
            npml = [layers if Nx > 2 * layers + 1 else 0] * 2 + [layers if Nz > 2 * layers + 1 else 0] * 2
            for mode, prec, disp in itertools.product((0, 1), (32, 64), (0, 1)):
                iJ, kJ = int(rng.integers(0, Nx)), int(rng.integers(0, Nz + 1))
                eng = fdtd_amd.Engine2D(Nx, Nz, npml, dx, dz, dt, K["mu0"], K["eps0"], iJ, kJ, steps, precision=prec, mode=mode)
                tw = po.Twin2D(prec, Nx, Nz, npml, dx, dz, dt, K["mu0"], K["eps0"], iJ, kJ, steps)
                mask = (rng.random((Nx, Nz)) < 0.4) if disp else np.zeros((Nx, Nz), dtype=bool)
                Cs = [np.where(mask, 0.02 * rng.random((Nx, Nz)), 0.0), np.where(mask, -1.9 + 0.01 * rng.random((Nx, Nz)), 0.0),
                      np.where(mask, 1 + 0.01 * rng.random((Nx, Nz)), 0.0)]
                jy = rng.standard_normal(steps) * 1e3
                dd = (dx, dz)
                for be in (eng, tw):
                    for a in range(2):
                        for s_ in range(2):
                            if npml[a * 2 + s_]:
                                tabs = po.cpml_face(npml[a * 2 + s_], s_, dd[a], dz, dt)
                                for k in range(6):
                                    be.set(po.D2_TAB + (a * 2 + s_) * 6 + k, tabs[k])
                    for q in range(3):
                        be.set(po.D2_C1 + q, Cs[q])
                    be.set(po.D2_JY, jy)
                eng.step(0, steps)
                tw.step(0, steps)
                what = f"2-D {Nx}x{Nz} npml={npml} fp{prec} mode {mode} dispersive={disp}"
                if mode == MODE2D_FUSED:
                    compare_2d(eng, tw.get, Nx, Nz, npml, mask, what)
                else:
                    for id_ in po.FIELDS2D + po.ALL_PHI2D:
                        if eng.size(id_):
                            util.assert_equal_arrays(eng.get(id_), tw.get(id_), f"{what}, array {id_}")
                eng.close()
                cases += 1
    assert cases == 256
