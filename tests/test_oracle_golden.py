"""The CPU twin against the golden vectors generated from the compiled reference (CPU only)."""
import json
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from tests import util

FIX3D = ["g1_3d_cpml", "g2_3d_cavity", "g3_3d_hetero"]


@pytest.mark.parametrize("name", FIX3D)
def test_twin64_matches_reference_3d(name):
    meta, ins, outs, extra = util.load_fixture(name)
    tw = util.twin3d_from_meta(meta, 64)
    util.feed(tw, ins)
    probe = []
    on_step = None
    if "probe" in extra:
        cell = tuple(meta["probe_cell"])
        shp = po.shape3d(po.EX, meta["N"], meta["npml"])
        on_step = lambda n: probe.append(tw.get(po.EX, shp)[cell])
    got = util.run_snapshots(tw, meta, outs, on_step)
    for key, want in outs.items():
        util.assert_equal_arrays(got[key], want, f"{name} step {key[0]} array {key[1]}")
    if probe:
        util.assert_equal_arrays(np.array(probe), extra["probe"], f"{name} probe series")


@pytest.mark.parametrize("name", FIX3D)
def test_twin32_close_to_reference_3d(name):
    """fp32 twin vs the fp64 reference: round-off drift only.  Stated tolerance, relative L2 of each
    vector field (all three components together) for up to 120 steps: D and E (what the reference
    writes out) <= 1e-6; B and H <= 3e-6 (the lossless cavity case sits at 1.9e-6)."""
    meta, ins, outs, _ = util.load_fixture(name)
    tw = util.twin3d_from_meta(meta, 32)
    util.feed(tw, ins)
    got = util.run_snapshots(tw, meta, outs)
    for step in meta["snapshots"]:
        for fam in range(4):          # D, E, B, H
            ids = [fam * 3 + c for c in range(3)]
            assert util.family_rel_l2([got[(step, i)] for i in ids], [outs[(step, i)] for i in ids]) < (1e-6 if fam < 2 else 3e-6), (name, step, fam)


def test_mu_grid_equal_to_mu0_reproduces_reference():
    meta, ins, outs, _ = util.load_fixture("g3_3d_hetero")
    tw = util.twin3d_from_meta(meta, 64, has_mu=True)
    util.feed(tw, ins)
    for c in range(3):
        tw.set(po.MU_X + c, np.full(tw.size(po.MU_X + c), meta["mu0"]))
    got = util.run_snapshots(tw, meta, outs)
    for key, want in outs.items():
        util.assert_equal_arrays(got[key], want, f"mu grid, array {key[1]}")


def test_twin64_matches_reference_2d_slab():
    meta, ins, outs, _ = util.load_fixture("g4_2d_slab")
    tw = util.twin2d_from_meta(meta, 64)
    util.feed(tw, ins)
    got = util.run_snapshots(tw, meta, outs)
    for key, want in outs.items():
        util.assert_equal_arrays(got[key], want, f"g4 array {key[1]}")


def test_twin32_close_to_reference_2d_slab():
    """Stated tolerance: relative L2 <= 5e-6 per field after 160 steps through the resonant slab."""
    meta, ins, outs, _ = util.load_fixture("g4_2d_slab")
    tw = util.twin2d_from_meta(meta, 32)
    util.feed(tw, ins)
    got = util.run_snapshots(tw, meta, outs)
    for (step, id_), want in outs.items():
        if id_ in (po.D2_DY, po.D2_EY, po.D2_BX, po.D2_BZ, po.D2_HX, po.D2_HZ):
            assert util.rel_l2(got[(step, id_)], want) < 5e-6, (step, id_)


def test_twin64_matches_reference_2d_pec512():
    """512x512, PEC walls, point source, 2000 steps on the CPU path."""
    meta, ins, outs, extra = util.load_fixture("g5_2d_pec512")
    tw = util.twin2d_from_meta(meta, 64)
    util.feed(tw, ins)
    shp = po.shape2d(po.D2_EY, meta["Nx"], meta["Nz"], meta["npml"])
    cell, sub = tuple(meta["probe_cell"]), meta["subsample"]
    probe = []
    n = 0
    for snap in meta["snapshots"]:
        for m in range(n, snap):
            tw.step(m, 1)
            probe.append(tw.get(po.D2_EY, shp)[cell])
        n = snap
        ey = tw.get(po.D2_EY, shp)
        util.assert_equal_arrays(ey[::sub, ::sub], outs[(snap, po.D2_EY)], f"g5 Ey at {snap}")
        assert util.seq_sumsq(ey) == meta["sumsq_ey"][str(snap)]
    util.assert_equal_arrays(np.array(probe), extra["probe"], "g5 probe series")


def test_setup_generators_match_fixture_inputs():
    """The restated CPML / source formulas give the tables the reference built (same libm)."""
    meta, ins, _, _ = util.load_fixture("g3_3d_hetero")
    d = (meta["dx"], meta["dy"], meta["dz"])
    for a in range(3):
        for s in range(2):
            tabs = po.cpml_face(meta["npml"][a * 2 + s], s, d[a], d[a], meta["dt"])
            for k in range(6):
                util.assert_equal_arrays(tabs[k], ins[po.tab_id(a, s, k)], f"table axis {a} side {s} kind {k}")
    wf, mag = po.default_source3d(meta["N"][0], meta["N"][1], meta["steps"], d, meta["dt"])
    util.assert_equal_arrays(wf, ins[po.JX_WF], "waveform")
    util.assert_equal_arrays(mag, ins[po.JX_MAG], "source magnitude")
    m2, i2, _, _ = util.load_fixture("g4_2d_slab")
    dd = (m2["dx"], m2["dz"])
    for a in range(2):
        for s in range(2):
            tabs = po.cpml_face(m2["npml"][a * 2 + s], s, dd[a], m2["dz"], m2["dt"])
            for k in range(6):
                util.assert_equal_arrays(tabs[k], i2[po.D2_TAB + (a * 2 + s) * 6 + k], f"2d table {a} {s} {k}")
    util.assert_equal_arrays(po.default_source2d(m2["steps"], m2["dx"], m2["dz"], m2["dt"]), i2[po.D2_JY], "Jy")
    c = po.default_lorentz2d(m2["Nx"], m2["Nz"], m2["dt"])
    for q in range(3):
        util.assert_equal_arrays(c[q], i2[po.D2_C1 + q], f"C{q + 1}")


def test_known_answers_on_reference_defaults():
    """Sum of squares after a few steps of the unmodified default problems."""
    with open(os.path.join(util.GOLDEN, "g6_known_answers.json")) as f:
        ka = json.load(f)
    K = po.consts()
    # 2-D default: f0 = 1.5e14, 40 cells per wavelength, 5 x 12 wavelengths, CFL 2, 20 layers
    lam = K["c0"] / 1.5e14
    dx = lam / 40
    Nx, Nz = int(round(5.0 * lam / dx)), int(round(12.0 * lam / dx))
    dt = dx / (K["c0"] * 2.0)
    assert (Nx, Nz) == (200, 480)
    npml = (20,) * 4
    tw = po.Twin2D(64, Nx, Nz, npml, dx, dx, dt, K["mu0"], K["eps0"], Nx // 2 - 40, Nz // 2, 1580)
    for a in range(2):
        for s in range(2):
            tabs = po.cpml_face(20, s, dx, dx, dt)
            for k in range(6):
                tw.set(po.D2_TAB + (a * 2 + s) * 6 + k, tabs[k])
    tw.set(po.D2_JY, po.default_source2d(1580, dx, dx, dt))
    for q, c in enumerate(po.default_lorentz2d(Nx, Nz, dt)):
        tw.set(po.D2_C1 + q, c)
    tw.step(0, 200)
    assert util.seq_sumsq(tw.get(po.D2_EY)) == ka["sumsq_ey_2d_default_200_steps"]
    # 3-D default: 1053 nm, 20 cells per wavelength, 10 wavelengths cube, 20 layers, plane k = 20
    lam = 1053e-9
    d = lam / 20
    N = (int(round(10.0 * lam / d)),) * 3
    dt = d / (K["c0"] * 2.0)
    assert N == (200, 200, 200)
    t3 = po.Twin3D(64, N, (20,) * 6, (d, d, d), dt, K["mu0"], 20, 1501)
    for a in range(3):
        for s in range(2):
            tabs = po.cpml_face(20, s, d, d, dt)
            for k in range(6):
                t3.set(po.tab_id(a, s, k), tabs[k])
    wf, mag = po.default_source3d(N[0], N[1], 1501, (d, d, d), dt)
    t3.set(po.JX_WF, wf)
    t3.set(po.JX_MAG, mag)
    for c, e in enumerate(po.default_materials3d(N, (d, d, d))):
        t3.set(po.EPS_X + c, e)
    t3.step(0, 5)
    assert util.seq_sumsq(t3.get(po.EX)) == ka["sumsq_ex_3d_default_5_steps"]


def test_twin64_matches_reference_2d_default_over_its_full_run():
    """The unmodified 2-D default over all 1580 steps (fixture g10): probe series of every step and
    the snapshots, bit for bit."""
    meta, ins, outs, extra = util.load_fixture("g10_2d_default_full_run")
    tw = util.twin2d_from_meta(meta, 64)
    util.feed(tw, ins)
    cells = [(id_, po.shape2d(id_, meta["Nx"], meta["Nz"], meta["npml"]), tuple(c)) for id_, c in meta["probes"]]
    series = np.zeros((meta["steps"], len(cells)))

    def on_step(n):
        for q, (id_, shp, c) in enumerate(cells):
            series[n, q] = tw.get(id_, shp)[c]

    got = util.run_snapshots(tw, meta, outs, on_step)
    for key, want in outs.items():
        util.assert_equal_arrays(got[key], want, f"g10 step {key[0]} array {key[1]}")
    util.assert_equal_arrays(series, extra["probe"], "g10 probe series")


def test_twin64_matches_reference_3d_default_first_200_steps():
    """The unmodified 3-D default (fixture g9, 200^3): the first 200 of its 1501 steps on the CPU twin --
    the whole run takes the twin several minutes; the GPU suite covers all 1501 steps.  eps comes from
    the host restatement of setupMaterials and must carry the reference's checksums."""
    from tests.golden.make_golden import checksum
    meta, ins, outs, extra = util.load_fixture("g9_3d_default_full_run")
    tw = util.twin3d_from_meta(meta, 64)
    util.feed(tw, ins)
    eps = po.default_materials3d(meta["N"], (meta["dx"], meta["dy"], meta["dz"]))
    for c in range(3):
        assert checksum(eps[c]) == meta["eps_checksums"][str(po.EPS_X + c)]
        tw.set(po.EPS_X + c, eps[c])
    snap, sub = 200, meta["subsample"]
    tw.step(0, snap)
    for id_ in po.FIELDS3D:
        full = tw.get(id_, po.shape3d(id_, meta["N"], meta["npml"]))
        assert checksum(full) == meta["checksums"][f"{snap}:{id_}"], f"g9 step {snap} array {id_}"
        if (snap, id_) in outs:
            util.assert_equal_arrays(full[::sub, ::sub, ::sub], outs[(snap, id_)], f"g9 step {snap} array {id_}")
