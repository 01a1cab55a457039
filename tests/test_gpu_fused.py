"""One-kernel-per-step (fused E+H) stepping against the golden vectors and the twin (needs a GPU)."""
import os

import numpy as np
import pytest

import fdtd_amd
from fdtd_amd.engine import MODE_FUSED
from oracle import pyoracle as po
from tests import util
from tests.test_gpu_parity import FIX3D, engine3d_from_meta

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", FIX3D)
@pytest.mark.parametrize("prec", [32, 64])
@pytest.mark.parametrize("rows,chunk", [(16, 8), (8, 3), (3, 1)])
def test_fused_step_equals_reference_3d(name, prec, rows, chunk):
    """fp64 against the reference vectors, fp32 against the fp32 twin, bit for bit, for several
    workgroup shapes (rows per group, planes per chunk) so that every halo path is crossed."""
    meta, ins, outs, _ = util.load_fixture(name)
    eng = engine3d_from_meta(meta, prec, mode=MODE_FUSED)
    eng.tune(chunk, 1, rows)
    util.feed(eng, ins)
    got = util.run_snapshots(eng, meta, outs)
    if prec == 64:
        ref = outs
    else:
        tw = util.twin3d_from_meta(meta, 32)
        util.feed(tw, ins)
        ref = util.run_snapshots(tw, meta, outs)
    for key in outs:
        util.assert_equal_arrays(got[key], ref[key], f"{name} fused fp{prec} step {key[0]} array {key[1]}")


@pytest.mark.parametrize("prec", [32, 64])
def test_fused_mu_grid_3d(prec):
    meta, ins, outs, _ = util.load_fixture("g3_3d_hetero")
    eng = engine3d_from_meta(meta, prec, has_mu=True, mode=MODE_FUSED)
    tw = util.twin3d_from_meta(meta, prec, has_mu=True)
    util.feed(eng, ins)
    util.feed(tw, ins)
    eng.fill_materials(77, meta["eps0"], 4 * meta["eps0"], meta["mu0"], 2 * meta["mu0"])
    tw.fill_materials(77, meta["eps0"], 4 * meta["eps0"], meta["mu0"], 2 * meta["mu0"])
    eng.step(0, meta["steps"])
    tw.step(0, meta["steps"])
    for id_ in po.FIELDS3D + po.ALL_PHI3D:
        util.assert_equal_arrays(eng.get(id_), tw.get(id_), f"fused mu grid array {id_}")


@pytest.mark.parametrize("prec,N,npml", [
    (32, (300, 9, 11), (5, 6, 2, 2, 3, 3)),       # two x tiles, plain columns at the tile boundary
    (32, (270, 20, 6), (3, 20, 1, 0, 0, 2)),      # x tile boundary (256) inside the high CPML strip
    (64, (140, 6, 19), (10, 20, 0, 3, 4, 4)),     # fp64 tiles are 128 columns: boundary inside the strip
    (32, (7, 6, 5), (0, 0, 0, 0, 0, 0)),
    (64, (12, 33, 9), (4, 4, 4, 4, 4, 4)),        # more rows than one workgroup holds
])
@pytest.mark.parametrize("rows,chunk", [(16, 8), (5, 2)])
def test_fused_ragged_shapes_3d(prec, N, npml, rows, chunk):
    K = po.consts()
    d = (5e-8, 6e-8, 7e-8)
    dt = 5e-8 / (2 * K["c0"])
    steps = 25
    kJ = N[2] // 2
    eng = fdtd_amd.Engine3D(N, npml, d, dt, K["mu0"], kJ, steps, precision=prec, mode=MODE_FUSED)
    eng.tune(chunk, 1, rows)
    tw = po.Twin3D(prec, N, npml, d, dt, K["mu0"], kJ, steps)
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
    # an odd number of steps first, so that reads after the run come from the second buffer
    eng.step(0, 11)
    tw.step(0, 11)
    for id_ in po.FIELDS3D + po.ALL_PHI3D:
        if eng.size(id_):
            util.assert_equal_arrays(eng.get(id_), tw.get(id_), f"N={N} step 11 array {id_}")
    eng.step(11, steps - 11)
    tw.step(11, steps - 11)
    for id_ in po.FIELDS3D + po.ALL_PHI3D:
        if eng.size(id_):
            util.assert_equal_arrays(eng.get(id_), tw.get(id_), f"N={N} array {id_}")


def run_fused_slabs(meta, ins, parts, prec, steps, chunk, has_mu=False, extra_feed=None, use_top=True):
    """Several FUSED slab engines in one process; ghost planes move by device copies exactly where
    the send/recv pairs sit in fdtd_amd.slab.FusedSlabStepper."""
    from fdtd_amd import engine as E
    engs = [fdtd_amd.Engine3D(meta["N"], meta["npml"], (meta["dx"], meta["dy"], meta["dz"]), meta["dt"], meta["mu0"],
                              meta["k_Jx"], meta["steps"], precision=prec, mode=MODE_FUSED, z_range=p, has_mu=has_mu) for p in parts]
    for e in engs:
        e.tune(chunk, 1, 0)
        util.feed(e, ins)
        if extra_feed:
            extra_feed(e)
    pl = [[{w: [E.halo_tensor(e, w, c, par) for c in (0, 1)] for w in range(4)} for par in (0, 1)] for e in engs]
    for n in range(steps):
        for e in engs:
            e.fused(n, E.PART_EDGE)
        cur = [pl[r][engs[r].parity()] for r in range(len(engs))]
        for r in range(1, len(engs)):
            for c in (0, 1):
                cur[r - 1][E.HALO_RECV_HI][c].copy_(cur[r][E.HALO_SEND_DOWN][c])
        if use_top:          # the stepper's order: top planes, top layer finished, bulk last
            for e in engs:
                e.fused(n, E.PART_TOP)
            for e in engs:
                e.fused_top()
            for e in engs:
                e.fused(n, E.PART_INTERIOR)
        else:
            for e in engs:
                e.fused(n, E.PART_INTERIOR)
            for e in engs:
                e.fused_top()
        for r in range(len(engs) - 1):
            for c in (0, 1):
                cur[r + 1][E.HALO_RECV_LO][c].copy_(cur[r][E.HALO_SEND_UP][c])
    return engs


@pytest.mark.parametrize("use_top", [True, False])
@pytest.mark.parametrize("name,world,chunk,edge_planes", [("g1_3d_cpml", 2, 8, 1), ("g3_3d_hetero", 3, 2, 2), ("g2_3d_cavity", 4, 3, 3),
                                                          ("g3_3d_hetero", 8, 16, 1), ("g1_3d_cpml", 3, 16, 8)])
def test_fused_slabs_reproduce_reference(name, world, chunk, edge_planes, use_top, monkeypatch):
    from fdtd_amd import slab
    from tests.test_gpu_slabs import assemble
    monkeypatch.setenv("FDTD_EDGE_PLANES", str(edge_planes))     # planes of the EDGE and TOP launches, read at creation
    meta, ins, outs, _ = util.load_fixture(name)
    steps = meta["steps"]
    engs = run_fused_slabs(meta, ins, slab.partition(meta["N"][2], world), 64, steps, chunk, use_top=use_top)
    for (step, id_), want in outs.items():
        if step == steps:
            util.assert_equal_arrays(assemble(engs, id_), want, f"{name} x{world} fused slabs, array {id_}")


def test_fused_slabs_fp32_mu_grid_equal_whole_grid():
    from tests.test_gpu_slabs import assemble
    meta, ins, outs, _ = util.load_fixture("g3_3d_hetero")
    K = meta
    fill = lambda e: e.fill_materials(5, K["eps0"], 3 * K["eps0"], K["mu0"], 2 * K["mu0"])
    whole = run_fused_slabs(meta, ins, [(0, meta["N"][2])], 32, 31, 4, has_mu=True, extra_feed=fill)
    parts = run_fused_slabs(meta, ins, [(0, 5), (5, 6), (6, 19), (19, 26)], 32, 31, 4, has_mu=True, extra_feed=fill)
    for id_ in po.FIELDS3D + po.ALL_PHI3D:
        util.assert_equal_arrays(assemble(parts, id_), assemble(whole, id_), f"uneven fused slabs, array {id_}")


def test_fused_part_order_is_checked():
    K = po.consts()
    eng = fdtd_amd.Engine3D((16, 16, 16), (0,) * 6, (1e-7,) * 3, 1e-16, K["mu0"], 4, 3, precision=32, mode=MODE_FUSED, z_range=(0, 8))
    with pytest.raises(fdtd_amd.engine.EngineError):
        eng.fused(0, fdtd_amd.engine.PART_INTERIOR)
    with pytest.raises(fdtd_amd.engine.EngineError):
        eng.fused(0, fdtd_amd.engine.PART_TOP)
    eng.fused(0, fdtd_amd.engine.PART_EDGE)
    with pytest.raises(fdtd_amd.engine.EngineError):
        eng.fused(0, fdtd_amd.engine.PART_EDGE)
    with pytest.raises(fdtd_amd.engine.EngineError):
        eng.fused_top()                                    # the top layer's part has not run yet
    eng.fused(0, fdtd_amd.engine.PART_TOP)
    eng.fused_top()
    with pytest.raises(fdtd_amd.engine.EngineError):
        eng.fused(0, fdtd_amd.engine.PART_TOP)
    eng.fused(0, fdtd_amd.engine.PART_INTERIOR)
    lean = fdtd_amd.Engine3D((16, 16, 16), (0,) * 6, (1e-7,) * 3, 1e-16, K["mu0"], 4, 3, precision=32, mode=fdtd_amd.engine.MODE_LEAN)
    with pytest.raises(fdtd_amd.engine.EngineError):
        lean.fused(0)


def test_probe_on_fused_slabs_matches_reference_series():
    """A probe lives on the slab that owns its layer; with the slab step order (lowest chunk, rest,
    top layer) its series must still be the reference's."""
    from fdtd_amd import engine as E
    from fdtd_amd import slab
    meta, ins, outs, extra = util.load_fixture("g2_3d_cavity")
    parts = slab.partition(meta["N"][2], 3)
    engs = [fdtd_amd.Engine3D(meta["N"], meta["npml"], (meta["dx"], meta["dy"], meta["dz"]), meta["dt"], meta["mu0"],
                              meta["k_Jx"], meta["steps"], precision=64, mode=MODE_FUSED, z_range=p) for p in parts]
    cell = tuple(meta["probe_cell"])
    owners = []
    for e, (lo, hi) in zip(engs, parts):
        e.tune(4, 1, 6)
        util.feed(e, ins)
        if lo <= cell[2] < hi:
            e.probe_add(po.EX, *cell)
            owners.append(e)
        else:
            with pytest.raises(E.EngineError):
                e.probe_add(po.EX, *cell)
    assert len(owners) == 1
    pl = [[{w: [E.halo_tensor(e, w, c, par) for c in (0, 1)] for w in range(4)} for par in (0, 1)] for e in engs]
    for n in range(meta["steps"]):
        for e in engs:
            e.fused(n, E.PART_EDGE)
        cur = [pl[r][engs[r].parity()] for r in range(3)]
        for r in range(1, 3):
            for c in (0, 1):
                cur[r - 1][E.HALO_RECV_HI][c].copy_(cur[r][E.HALO_SEND_DOWN][c])
        for e in engs:
            e.fused(n, E.PART_INTERIOR)
            e.fused_top()
            e.probe_record()
        for r in range(2):
            for c in (0, 1):
                cur[r + 1][E.HALO_RECV_LO][c].copy_(cur[r][E.HALO_SEND_UP][c])
    util.assert_equal_arrays(owners[0].probe_read()[:, 0], extra["probe"], "probe series on fused slabs")


def _random_case(rng):
    N = tuple(int(v) for v in rng.integers(4, 41, size=3))
    if rng.random() < 0.3:
        N = (int(rng.integers(250, 300)), N[1], N[2])             # more than one x tile
    npml = []
    for a in range(3):
        lo, hi = (int(rng.integers(0, 7)) if rng.random() < 0.7 else 0 for _ in range(2))
        while lo + hi >= N[a] - 1:
            lo, hi = lo // 2, hi // 2
        npml += [lo, hi]
    if npml[0] and npml[1] and (N[0] - npml[1]) // 4 * 4 < (npml[0] + 1 + 3) // 4 * 4:
        npml[1] = 0                                                 # the two x strips must stay disjoint
    return N, tuple(npml), int(rng.choice([32, 64])), bool(rng.random() < 0.5), int(rng.integers(2, 17)), int(rng.integers(1, 12))


@pytest.mark.parametrize("seed", range(24))
def test_fused_random_configurations_equal_twin(seed):
    """Random grid shapes, CPML layer counts (zero included, per face), precision, mu grid and
    workgroup shape: the fused kernel must agree with the CPU twin bit for bit on every field and
    every CPML auxiliary array."""
    rng = np.random.default_rng(1000 + seed)
    N, npml, prec, has_mu, rows, chunk = _random_case(rng)
    K = po.consts()
    d = (5e-8, 6e-8, 7e-8)
    dt = 5e-8 / (2 * K["c0"])
    steps = 13
    kJ = int(rng.integers(0, N[2] + 1))
    narrow = seed % 3 == 2               # every third case: half-width lanes (8 bytes per lane instead of 16)
    if narrow:
        os.environ["FDTD_FUSED_VEC"] = "2" if prec == 32 else "1"
    try:
        eng = fdtd_amd.Engine3D(N, npml, d, dt, K["mu0"], kJ, steps, precision=prec, mode=MODE_FUSED, has_mu=has_mu)
    finally:
        os.environ.pop("FDTD_FUSED_VEC", None)
    order = (int(rng.integers(0, 2)), 2, 3, 5, 65536)[seed % 5]      # tile order over the XCDs: dealt / contiguous / runs / automatic
    eng.tune(chunk, order, rows)
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
            util.assert_equal_arrays(eng.get(id_), tw.get(id_), f"seed {seed}: N={N} npml={npml} fp{prec} mu={has_mu} rows={rows} chunk={chunk}, array {id_}")


@pytest.mark.parametrize("seed", range(10))
def test_fused_random_slab_partitions_equal_twin(seed):
    """Random uneven z partitions (single-layer slabs included) of random small problems, stepped in the
    slab order EDGE, TOP, top layer, INTERIOR with device copies for the exchanges: == the twin."""
    from fdtd_amd import engine as E
    from tests.test_gpu_slabs import assemble
    rng = np.random.default_rng(7000 + seed)
    N, npml, prec, has_mu, rows, chunk = _random_case(rng)
    N = (N[0], N[1], max(N[2], 12))
    cuts = sorted(set(int(v) for v in rng.integers(1, N[2], size=int(rng.integers(1, 4)))))
    parts = list(zip([0] + cuts, cuts + [N[2]]))
    K = po.consts()
    d = (5e-8, 6e-8, 7e-8)
    dt = 5e-8 / (2 * K["c0"])
    steps = 9
    kJ = int(rng.integers(0, N[2] + 1))
    engs = [fdtd_amd.Engine3D(N, npml, d, dt, K["mu0"], kJ, steps, precision=prec, mode=MODE_FUSED, has_mu=has_mu, z_range=p) for p in parts]
    tw = po.Twin3D(prec, N, npml, d, dt, K["mu0"], kJ, steps, has_mu)
    wf = rng.standard_normal(steps)
    mag = rng.random((N[0], N[1] + 1))
    for be in engs + [tw]:
        for a in range(3):
            for s in range(2):
                if npml[a * 2 + s]:
                    tabs = po.cpml_face(npml[a * 2 + s], s, d[a], d[a], dt)
                    for k in range(6):
                        be.set(po.tab_id(a, s, k), tabs[k])
        be.set(po.JX_WF, wf)
        be.set(po.JX_MAG, mag)
        be.fill_materials(seed, K["eps0"], 4 * K["eps0"], K["mu0"], 3 * K["mu0"])
    for e in engs:
        e.tune(chunk, 1, rows)
    pl = [[{w: [E.halo_tensor(e, w, c, par) for c in (0, 1)] for w in range(4)} for par in (0, 1)] for e in engs]
    R = len(engs)
    for n in range(steps):
        for e in engs:
            e.fused(n, E.PART_EDGE)
        cur = [pl[r][engs[r].parity()] for r in range(R)]
        for r in range(1, R):
            for c in (0, 1):
                cur[r - 1][E.HALO_RECV_HI][c].copy_(cur[r][E.HALO_SEND_DOWN][c])
        for e in engs:
            e.fused(n, E.PART_TOP)
        for e in engs:
            e.fused_top()
        for e in engs:
            e.fused(n, E.PART_INTERIOR)
        for r in range(R - 1):
            for c in (0, 1):
                cur[r + 1][E.HALO_RECV_LO][c].copy_(cur[r][E.HALO_SEND_UP][c])
    tw.step(0, steps)
    for id_ in po.FIELDS3D + po.ALL_PHI3D:
        if engs[0].size(id_):
            util.assert_equal_arrays(assemble(engs, id_), tw.get(id_), f"seed {seed}: N={N} npml={npml} parts={parts} fp{prec} mu={has_mu}, array {id_}")
