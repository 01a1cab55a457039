"""Parity over the reference's own run lengths (needs a GPU).

The reference runs its 3-D default for 1501 steps (FDTD3D.cpp:39) and its 2-D default for 1580
(FDTD2D.cpp:38).  The fixtures g9 / g10 were taken from the compiled, unmodified reference over those
runs: a probe series at several cells for every step, snapshots at three steps, and -- in 3-D --
left-to-right sums of squares and whole-array checksums of all twelve field arrays.

fp64 engine: bit for bit on all of it, in every storage mode.
fp32 engine: relative L2 distance from the reference per vector field, against a tolerance that is
stated as a function of the step (TOL3D / TOL2D below; the measured curves are in
profiles/r2_fp32_drift.md).  Where that exceeds 1e-6, `precision = 64` is the bit-exact mode.
"""
import numpy as np
import pytest

import fdtd_amd
from fdtd_amd import engine as E
from oracle import pyoracle as po
from tests import util

pytestmark = pytest.mark.gpu


def _engine3d(meta, ins, prec, mode):
    eng = fdtd_amd.Engine3D(meta["N"], meta["npml"], (meta["dx"], meta["dy"], meta["dz"]), meta["dt"], meta["mu0"],
                            meta["k_Jx"], meta["steps"], precision=prec, mode=mode)
    util.feed(eng, ins)
    eng.fill_reference_materials()            # FDTD3D::setupMaterials on the device, in fp64
    for id_, cell in meta["probes"]:
        eng.probe_add(id_, *cell)
    return eng


@pytest.mark.parametrize("mode", [E.MODE_FUSED, E.MODE_FULL, E.MODE_LEAN])
def test_3d_default_1501_steps_fp64_bit_exact(mode):
    meta, ins, outs, extra = util.load_fixture("g9_3d_default_full_run")
    eng = _engine3d(meta, ins, 64, mode)
    for c in range(3):                        # the regenerated eps is the reference's, every bit of it
        assert eng.checksum(E.EPS_X + c) == meta["eps_checksums_without_wall_column"][str(E.EPS_X + c)]
    resident = {E.MODE_FULL: po.FIELDS3D, E.MODE_LEAN: (0, 1, 2, 6, 7, 8), E.MODE_FUSED: (0, 1, 2, 6, 7, 8)}[mode]
    n, sub = 0, meta["subsample"]
    for snap in meta["snapshots"]:
        eng.step(n, snap - n)
        n = snap
        for id_ in resident:                  # all 8e6 cells of every resident array, without a download
            assert eng.checksum(id_) == meta["checksums"][f"{snap}:{id_}"], f"mode {mode} step {snap} array {id_}"
        for (step, id_), want in outs.items():
            if step != snap:
                continue
            full = eng.get(id_, po.shape3d(id_, meta["N"], meta["npml"]))
            util.assert_equal_arrays(full[::sub, ::sub, ::sub], want, f"mode {mode} step {snap} array {id_} (every {sub}th cell)")
            assert util.seq_sumsq(full) == meta["sumsq"][f"{snap}:{id_}"]
    util.assert_equal_arrays(eng.probe_read(), extra["probe"], f"mode {mode}: probe series over {meta['steps']} steps")


# fp32 against the fp64 reference, 3-D default problem: round-off accumulates like a random walk while the source
# feeds the grid and the CPML drains it.  Measured (profiles/r2_fp32_drift.md): 3.1e-7 after 200 steps, 1.0e-6 after
# 700, 1.54e-6 after 1501, i.e. 4e-8 * sqrt(steps); the tolerance is 1.5 times that curve.  It crosses the 1e-6 of
# the specification near step 280 (the measured curve near step 700): beyond that, precision = 64 is the mode
# that matches the reference (bit for bit, test above).
def TOL3D(step):
    return 6.0e-8 * max(step, 1) ** 0.5


def test_3d_default_1501_steps_fp32_drift_within_stated_curve():
    meta, ins, outs, extra = util.load_fixture("g9_3d_default_full_run")
    eng = _engine3d(meta, ins, 32, E.MODE_FUSED)
    n, sub = 0, meta["subsample"]
    report = []
    for snap in meta["snapshots"]:
        eng.step(n, snap - n)
        n = snap
        got = {id_: eng.get(id_, po.shape3d(id_, meta["N"], meta["npml"]))[::sub, ::sub, ::sub] for id_ in (3, 4, 5, 9, 10, 11)}
        e = util.family_rel_l2([got[i] for i in (3, 4, 5)], [outs[(snap, i)] for i in (3, 4, 5)])
        h = util.family_rel_l2([got[i] for i in (9, 10, 11)], [outs[(snap, i)] for i in (9, 10, 11)])
        report.append((snap, e, h))
        assert e <= TOL3D(snap) and h <= TOL3D(snap), f"fp32 drift at step {snap}: E {e:.2e}, H {h:.2e}, tolerance {TOL3D(snap):.2e}"
    series, want = eng.probe_read(), extra["probe"]
    # per probe, relative to the largest sample any probe of the same family (E or H) ever sees: the cross-polarised
    # components Ey, Ez are orders of magnitude below Ex, and what perturbs them is the round-off of Ex
    is_e = np.array([id_ < 6 for id_, _ in meta["probes"]])
    scale = np.where(is_e, np.abs(want[:, is_e]).max(), np.abs(want[:, ~is_e]).max())
    worst = (np.abs(series - want) / scale).max(axis=0)
    assert np.all(worst <= 4 * TOL3D(meta["steps"])), f"probe series: {worst}"      # pointwise maximum, not an L2 ratio
    print("fp32 vs reference, 3-D default:", [(s, f"{e:.2e}", f"{h:.2e}") for s, e, h in report], "probes", worst)


def _engine2d(meta, ins, prec, mode):
    eng = fdtd_amd.Engine2D(meta["Nx"], meta["Nz"], meta["npml"], meta["dx"], meta["dz"], meta["dt"], meta["mu0"], meta["eps0"],
                            meta["i_Jy"], meta["k_Jy"], meta["steps"], precision=prec, mode=mode)
    util.feed(eng, ins)
    for id_, cell in meta["probes"]:
        eng.probe_add(id_, *cell)
    return eng


@pytest.mark.parametrize("mode", [E.MODE2D_FUSED, E.MODE2D_FULL])
def test_2d_default_1580_steps_fp64_bit_exact(mode):
    meta, ins, outs, extra = util.load_fixture("g10_2d_default_full_run")
    eng = _engine2d(meta, ins, 64, mode)
    Nx, Nz = meta["Nx"], meta["Nz"]
    dispersive = np.asarray(ins[po.D2_C1]).reshape(Nx, Nz) != 0
    n = 0
    for snap in meta["snapshots"]:
        eng.step(n, snap - n)
        n = snap
        for (step, id_), want in outs.items():
            if step != snap:
                continue
            shp = po.shape2d(id_, Nx, Nz, meta["npml"])
            got, want = eng.get(id_, shp), want.reshape(shp)
            if id_ == po.D2_SEY and mode == E.MODE2D_FUSED:       # polarisation sums only exist at dispersive cells in this mode
                got, want = got[:Nx, :Nz][dispersive], want[:Nx, :Nz][dispersive]
            else:
                assert util.seq_sumsq(got) == meta["sumsq"][f"{snap}:{id_}"]
            util.assert_equal_arrays(got, want, f"2-D mode {mode} step {snap} array {id_}")
    util.assert_equal_arrays(eng.probe_read(), extra["probe"], f"2-D mode {mode}: probe series over {meta['steps']} steps")


# fp32 against the fp64 reference, 2-D default problem.  The slab is a LOSSLESS Lorentz resonator (delta = 0,
# FDTD2D.cpp:94-96) driven at the frequency where eps = mu = -1: the recursion's second-order difference
# (coefficients close to -2 and +1) integrates every rounding error twice, so fp32 drifts quadratically here, not
# like a random walk.  Measured (profiles/r2_fp32_drift.md): 1.3e-6 after 160 steps, 1.4e-5 after 400, 1.0e-4 after
# 1000, 1.7e-4 after 1580.  Stated tolerance: 1e-6 + 2e-10 * steps^2.  The 1e-6 of the specification holds for
# about the first 150 steps of this problem in fp32; the whole run needs precision = 64 (bit for bit, test above),
# which is what the fdtd2d driver uses by default.
def TOL2D(step):
    return 1.0e-6 + 2.0e-10 * step ** 2


@pytest.mark.parametrize("mode", [E.MODE2D_FUSED, E.MODE2D_FULL])
def test_2d_default_1580_steps_fp32_drift_within_stated_curve(mode):
    meta, ins, outs, extra = util.load_fixture("g10_2d_default_full_run")
    eng = _engine2d(meta, ins, 32, mode)
    n, report = 0, []
    for snap in meta["snapshots"]:
        eng.step(n, snap - n)
        n = snap
        ey = util.rel_l2(eng.get(po.D2_EY), outs[(snap, po.D2_EY)])
        h = util.family_rel_l2([eng.get(i) for i in (po.D2_HX, po.D2_HZ)], [outs[(snap, i)] for i in (po.D2_HX, po.D2_HZ)])
        report.append((snap, ey, h))
        assert ey <= TOL2D(snap) and h <= TOL2D(snap), f"fp32 drift at step {snap}: Ey {ey:.2e}, H {h:.2e}, tolerance {TOL2D(snap):.2e}"
    series, want = eng.probe_read(), extra["probe"]
    is_e = np.array([id_ == po.D2_EY for id_, _ in meta["probes"]])
    scale = np.where(is_e, np.abs(want[:, is_e]).max(), np.abs(want[:, ~is_e]).max())
    worst = (np.abs(series - want) / scale).max(axis=0)
    assert np.all(worst <= 4 * TOL2D(meta["steps"])), f"probe series: {worst}"      # pointwise maximum, not an L2 ratio
    print("fp32 vs reference, 2-D default:", [(s, f"{a:.2e}", f"{b:.2e}") for s, a, b in report], "probes", worst)
