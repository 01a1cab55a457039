"""fdtd3d_step_observed: the loop body of the reference's run() -- step, then advance the RMS window of the
cell-centred |E|^2 (FDTD3D.cpp:1085-1103) -- with the window advanced INSIDE the two-step launches.  The window
(the one being accumulated and the last completed one) must equal, bit for bit, what stepping one step at a
time and accumulating after each gives on the CPU twin (needs a GPU)."""
import numpy as np
import pytest

import fdtd_amd
from fdtd_amd.engine import MODE_FULL, MODE_FUSED, MODE_LEAN
from oracle import pyoracle as po
from tests import util
from tests.test_gpu_parity import FIX3D, engine3d_from_meta

pytestmark = pytest.mark.gpu


def _twin_windows(tw, n0, count, weight, window, kept):
    """Steps the twin as FDTD3D::run does; returns (sqrt of the window being accumulated, sqrt of the last completed one)."""
    for n in range(n0, n0 + count):
        tw.step(n, 1)
        tw.observe_accumulate(weight, n % window == 0)
        if (n + 1) % window == 0:
            kept = tw.observe_rms().copy()
    return tw.observe_rms(), kept


@pytest.mark.parametrize("name", FIX3D)
@pytest.mark.parametrize("prec", [32, 64])
@pytest.mark.parametrize("rows,chunk", [(16, 48), (6, 2), (9, 5)])
def test_window_inside_two_step_launches_equals_twin(name, prec, rows, chunk, monkeypatch):
    """Windows of 5 steps (they open and complete on first and on second steps of a launch alike), calls of odd and
    even length starting at odd and even steps, three workgroup shapes that put every halo row, column and plane
    somewhere inside the grid."""
    monkeypatch.setenv("FDTD_PAIR", "1")          # these grids are far below the size where fdtd3d_step pairs by itself
    meta, ins, _, _ = util.load_fixture(name)
    eng = engine3d_from_meta(meta, prec, mode=MODE_FUSED)
    assert eng.pair_info()["used_by_step"]
    eng.pair_shape(chunk, rows)
    tw = util.twin3d_from_meta(meta, prec)
    util.feed(eng, ins)
    util.feed(tw, ins)
    weight, window, kept, n = 0.1, 5, None, 0
    for count in (1, 6, 7, 2, 9):
        eng.step_observed(n, count, weight, window)
        cur, kept = _twin_windows(tw, n, count, weight, window, kept)
        n += count
        if n % window != 0:        # (right after a window has completed, the accumulator's content is undefined)
            util.assert_equal_arrays(eng.observe_rms(), cur, f"{name} fp{prec}: window being accumulated after step {n - 1}")
        if kept is not None:
            util.assert_equal_arrays(eng.observe_rms_window(), kept, f"{name} fp{prec}: last completed window after step {n - 1}")
    assert kept is not None and kept.max() > 0
    for id_ in po.FIELDS3D:
        util.assert_equal_arrays(eng.get(id_), tw.get(id_), f"{name} fp{prec}: array {id_} after {n} observed steps")


@pytest.mark.parametrize("prec,N,npml", [
    (32, (300, 20, 11), (5, 6, 2, 2, 3, 3)),      # three x tiles; both CPML strips in x tile 0
    (32, (256, 7, 9), (20, 20, 2, 2, 2, 2)),      # Nx = row pitch: the column right of the last cell is the wall, not padding
    (32, (130, 40, 9), (20, 20, 20, 16, 4, 4)),   # more rows than a workgroup owns
    (64, (140, 6, 19), (10, 8, 0, 3, 4, 4)),      # fp64 tiles own 60 columns
    (32, (64, 12, 40), (0, 0, 0, 0, 0, 0)),       # no CPML: the straight-line loop
])
@pytest.mark.parametrize("has_mu", [False, True])
@pytest.mark.parametrize("window", [1, 2, 4, 7])
def test_window_on_ragged_grids(prec, N, npml, has_mu, window, monkeypatch):
    monkeypatch.setenv("FDTD_PAIR", "1")
    K = po.consts()
    d = (5e-8, 6e-8, 7e-8)
    dt = 5e-8 / (2 * K["c0"])
    steps, kJ = 16, N[2] // 2
    rng = np.random.default_rng(29)
    wf, mag = po.default_source3d(N[0], N[1], steps, d, dt)
    mag = rng.random(mag.shape)
    eps = [K["eps0"] * (1 + 3 * rng.random(po.shape3d(po.EPS_X + c, N, npml))) for c in range(3)]
    mu = [K["mu0"] * (1 + rng.random(po.shape3d(po.MU_X + c, N, npml))) for c in range(3)]
    tw = po.Twin3D(prec, N, npml, d, dt, K["mu0"], kJ, steps, has_mu)
    eng = fdtd_amd.Engine3D(N, npml, d, dt, K["mu0"], kJ, steps, precision=prec, mode=MODE_FUSED, has_mu=has_mu)
    eng.pair_shape(4, 11)
    for be in (tw, eng):
        for a in range(3):
            for s in range(2):
                tabs = po.cpml_face(npml[a * 2 + s], s, d[a], d[a], dt)
                for k in range(6):
                    if npml[a * 2 + s]:
                        be.set(po.tab_id(a, s, k), tabs[k])
        be.set(po.JX_WF, wf)
        be.set(po.JX_MAG, mag)
        for c in range(3):
            be.set(po.EPS_X + c, eps[c])
            if has_mu:
                be.set(po.MU_X + c, mu[c])
    kept, n = None, 0
    for count in (3, 8, 5):
        eng.step_observed(n, count, 0.25, window)
        cur, kept = _twin_windows(tw, n, count, 0.25, window, kept)
        n += count
        if n % window != 0:
            util.assert_equal_arrays(eng.observe_rms(), cur, f"N={N} window {window}: accumulating, after step {n - 1}")
        util.assert_equal_arrays(eng.observe_rms_window(), kept, f"N={N} window {window}: completed, after step {n - 1}") if kept is not None else None
    assert kept is not None and kept.max() > 0


@pytest.mark.parametrize("mode", [MODE_FULL, MODE_LEAN, MODE_FUSED])
def test_observed_steps_without_two_step_launches(mode, monkeypatch):
    """Engines that step one step per launch (two-kernel storage modes, small grids) take the same call: step, then
    the observe kernel, and the completed window is kept the same way."""
    monkeypatch.setenv("FDTD_PAIR", "0")
    meta, ins, _, _ = util.load_fixture("g3_3d_hetero")
    eng = engine3d_from_meta(meta, 64, mode=mode)
    tw = util.twin3d_from_meta(meta, 64)
    util.feed(eng, ins)
    util.feed(tw, ins)
    eng.step_observed(0, 11, 0.1, 4)
    cur, kept = _twin_windows(tw, 0, 11, 0.1, 4, None)
    util.assert_equal_arrays(eng.observe_rms(), cur, "window being accumulated")
    util.assert_equal_arrays(eng.observe_rms_window(), kept, "last completed window")


def test_observed_refusals():
    meta, ins, _, _ = util.load_fixture("g1_3d_cpml")
    eng = engine3d_from_meta(meta, 32, mode=MODE_FUSED)
    with pytest.raises(fdtd_amd.engine.EngineError, match="no window has been completed"):
        eng.observe_rms_window()
    with pytest.raises(fdtd_amd.engine.EngineError, match="window must have at least one step"):
        eng.step_observed(0, 2, 0.1, 0)
    slab = fdtd_amd.Engine3D(meta["N"], meta["npml"], (meta["dx"], meta["dy"], meta["dz"]), meta["dt"], meta["mu0"], meta["k_Jx"],
                             meta["steps"], precision=32, mode=MODE_FUSED, z_range=(0, meta["N"][2] // 2))
    with pytest.raises(fdtd_amd.engine.EngineError, match="whole grid"):
        slab.step_observed(0, 2, 0.1, 4)


@pytest.mark.parametrize("seed", range(16))
def test_window_random_configurations(seed, monkeypatch=None):
    """Random grids (one to three x tiles), CPML layers per face, mu grid or not, precision, workgroup shape, window
    length and run lengths: fdtd3d_step_observed with two steps per launch against an engine that steps and
    accumulates one step at a time -- both windows and all six stored arrays, bit for bit."""
    import os
    os.environ["FDTD_PAIR"] = "1"
    try:
        rng = np.random.default_rng(5000 + seed)
        prec = int(rng.choice([32, 64]))
        tile = 124 if prec == 32 else 60
        N = (int(rng.choice([rng.integers(5, 40), tile - 1, tile, tile + 1, rng.integers(tile, 3 * tile)])),
             int(rng.integers(4, 46)), int(rng.integers(4, 40)))

        def layers(n):
            lo, hi = int(rng.integers(0, 1 + min(12, (n - 2) // 2))), int(rng.integers(0, 1 + min(12, (n - 2) // 2)))
            return (lo if rng.random() < 0.8 else 0, hi if rng.random() < 0.8 else 0)
        npml = layers(N[0]) + layers(N[1]) + layers(N[2])
        if npml[0] and npml[1] and (N[0] - npml[1]) // 4 * 4 < -(-(npml[0] + 1) // 4) * 4:
            npml = (0,) + npml[1:]
        has_mu = bool(rng.integers(0, 2))
        rows, chunk = int(rng.integers(6, 17)), int(rng.choice([1, 2, 3, 5, 8, 48]))
        window = int(rng.choice([1, 2, 3, 4, 5, 8]))
        K = po.consts()
        d = (5e-8, 6e-8, 7e-8)
        dt = 5e-8 / (2 * K["c0"])
        counts = [int(rng.integers(1, 8)) for _ in range(3)]
        steps = sum(counts)
        kJ = int(rng.integers(0, N[2] + 1))
        wf, mag = po.default_source3d(N[0], N[1], steps, d, dt)
        mag = rng.random(mag.shape)
        engs = [fdtd_amd.Engine3D(N, npml, d, dt, K["mu0"], kJ, steps, precision=prec, mode=MODE_FUSED, has_mu=has_mu) for _ in range(2)]
        engs[0].pair_shape(chunk, rows)
        for eng in engs:
            for a in range(3):
                for s_ in range(2):
                    if npml[a * 2 + s_]:
                        tabs = po.cpml_face(npml[a * 2 + s_], s_, d[a], d[a], dt)
                        for k in range(6):
                            eng.set(po.tab_id(a, s_, k), tabs[k])
            eng.set(po.JX_WF, wf)
            eng.set(po.JX_MAG, mag)
            eng.fill_materials(91 + seed, K["eps0"], 4 * K["eps0"], K["mu0"], (2 if has_mu else 1) * K["mu0"])
        what = f"N={N} npml={npml} mu={has_mu} fp{prec} rows={rows} chunk={chunk} window={window} counts={counts}"
        n, kept = 0, None
        for count in counts:
            engs[0].step_observed(n, count, 0.3, window)
            for m in range(n, n + count):
                engs[1].fused(m)
                engs[1].observe_accumulate(0.3, m % window == 0)
                if (m + 1) % window == 0:
                    kept = engs[1].observe_rms().copy()
            n += count
            if n % window != 0:
                util.assert_equal_arrays(engs[0].observe_rms(), engs[1].observe_rms(), f"{what}: accumulating after step {n - 1}")
            if kept is not None:
                util.assert_equal_arrays(engs[0].observe_rms_window(), kept, f"{what}: completed window after step {n - 1}")
        for id_ in [po.DX, po.DY, po.DZ, po.BX, po.BY, po.BZ]:
            util.assert_equal_arrays(engs[0].get(id_), engs[1].get(id_), f"{what} array {id_}")
    finally:
        os.environ.pop("FDTD_PAIR", None)
