"""Slab groups (fdtd3d_group_*): several z-slabs of one problem stepped from one host thread, ghost
planes copied device to device on streams.  On a one-GPU box all slabs sit on device 0; the
choreography, the events and the copies are the same ones a multi-GPU node runs (there the copies
are peer copies).  Every result must equal the whole-grid result bit for bit."""
import ctypes as C

import numpy as np
import pytest

import fdtd_amd
from fdtd_amd import engine as E
from oracle import pyoracle as po
from tests import util
from tests.test_gpu_parity import FIX3D

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[None, "1"], ids=["one-step", "two-step"])
def pair_env(request, monkeypatch):
    """Default: grids this small step one step per launch.  FDTD_PAIR=1: FUSED groups whose cuts allow it take two
    steps per launch on overlapping slabs (and single engines pair their steps too)."""
    if request.param is not None:
        monkeypatch.setenv("FDTD_PAIR", request.param)
    return request.param


def group_from_meta(meta, prec, n_slabs, mode, **kw):
    return E.EngineGroup3D(meta["N"], meta["npml"], (meta["dx"], meta["dy"], meta["dz"]), meta["dt"], meta["mu0"],
                           meta["k_Jx"], meta["steps"], n_slabs, precision=prec, mode=mode, **kw)


@pytest.mark.parametrize("name", FIX3D)
@pytest.mark.parametrize("mode", [E.MODE_FULL, E.MODE_LEAN, E.MODE_FUSED])
@pytest.mark.parametrize("n_slabs", [2, 3])
def test_group_fp64_equals_reference_golden(name, mode, n_slabs, pair_env):
    meta, ins, outs, _ = util.load_fixture(name)
    g = group_from_meta(meta, 64, n_slabs, mode)
    if name in ("g1_3d_cpml", "g2_3d_cavity"):           # (their cuts are clear of the z layers)
        assert g.pairs == (pair_env == "1" and mode == E.MODE_FUSED)
    util.feed(g, ins)
    got = util.run_snapshots(g, meta, outs)
    for key, want in outs.items():
        util.assert_equal_arrays(got[key], want, f"{name} group x{n_slabs} mode {mode} step {key[0]} array {key[1]}")
    g.close()


@pytest.mark.parametrize("mode", [E.MODE_LEAN, E.MODE_FUSED])
def test_group_fp32_equals_twin_and_single_engine(mode, pair_env):
    meta, ins, outs, _ = util.load_fixture("g3_3d_hetero")
    g = group_from_meta(meta, 32, 4, mode)
    tw = util.twin3d_from_meta(meta, 32)
    util.feed(g, ins)
    util.feed(tw, ins)
    got = util.run_snapshots(g, meta, outs)
    ref = util.run_snapshots(tw, meta, outs)
    for key in outs:
        util.assert_equal_arrays(got[key], ref[key], f"group fp32 mode {mode} step {key[0]} array {key[1]}")
    one = fdtd_amd.Engine3D(meta["N"], meta["npml"], (meta["dx"], meta["dy"], meta["dz"]), meta["dt"], meta["mu0"], meta["k_Jx"],
                            meta["steps"], precision=32, mode=mode)
    util.feed(one, ins)
    one.step(0, max(meta["snapshots"]))
    for id_ in (E.DX, E.DY, E.BZ):
        assert g.checksum(id_) == one.checksum(id_)
        assert g.sumsq(id_) == pytest.approx(one.sumsq(id_), rel=1e-12)


def test_group_observables_and_probe(pair_env):
    meta, ins, outs, extra = util.load_fixture("g2_3d_cavity")
    g = group_from_meta(meta, 64, 3, E.MODE_FUSED)
    tw = util.twin3d_from_meta(meta, 64)
    util.feed(g, ins)
    util.feed(tw, ins)
    pi, pj, pk = meta["probe_cell"]
    g.owner(pk).probe_add(E.EX, pi, pj, pk)
    for n in range(25):
        g.step(n, 1)
        tw.step(n, 1)
        g.observe_accumulate(0.1, n % 10 == 0)
        tw.observe_accumulate(0.1, n % 10 == 0)
    util.assert_equal_arrays(g.observe_rms(), tw.observe_rms(), "group RMS window")
    a, b = g.observe_frame()
    ta, tb = tw.observe_frame()
    util.assert_equal_arrays(a, ta, "group centred Ex")
    util.assert_equal_arrays(b, tb, "group |E|")
    util.assert_equal_arrays(g.owner(pk).probe_read()[:, 0], extra["probe"][:25], "group probe series")


@pytest.mark.parametrize("prec,N,npml", [
    (32, (300, 33, 44), (0,) * 6),                 # three x tiles of 124 columns, no CPML: the loop for waves strictly inside the grid
    (32, (390, 45, 50), (6, 6, 5, 5, 4, 4)),       # with CPML: the workgroups clear of the layers run that loop
    (64, (150, 33, 40), (0, 3, 2, 0, 0, 0)),       # fp64 tiles own 60 columns
])
@pytest.mark.parametrize("observed", [False, True])
def test_group_two_steps_per_launch_on_grids_of_several_x_tiles(prec, N, npml, observed, monkeypatch):
    """Overlapping slabs whose workgroups include waves strictly inside the grid in x, y and (globally) z: the top chunk
    of a slab below another one ends at its last stored plane, which such a wave must not read past."""
    monkeypatch.setenv("FDTD_PAIR", "1")
    K = po.consts()
    d = (5e-8, 6e-8, 7e-8)
    dt = 5e-8 / (2 * K["c0"])
    steps, kJ = 9, N[2] // 2 + 1
    g = E.EngineGroup3D(N, npml, d, dt, K["mu0"], kJ, steps, 3, precision=prec, has_mu=True, mode=E.MODE_FUSED)
    assert g.pairs
    for m in g.members:
        m.pair_shape(7, 9)
    tw = po.Twin3D(prec, N, npml, d, dt, K["mu0"], kJ, steps, True)
    rng = np.random.default_rng(77)
    wf, mag = rng.standard_normal(steps), rng.random((N[0], N[1] + 1))
    for be in (g, tw):
        for a in range(3):
            for s in range(2):
                if npml[a * 2 + s]:
                    tabs = po.cpml_face(npml[a * 2 + s], s, d[a], d[a], dt)
                    for k in range(6):
                        be.set(po.tab_id(a, s, k), tabs[k])
        be.set(po.JX_WF, wf)
        be.set(po.JX_MAG, mag)
        be.fill_materials(5, K["eps0"], 4 * K["eps0"], K["mu0"], 3 * K["mu0"])
    if observed:
        g.step_observed(0, steps, 0.2, 4)
        kept = None
        for n in range(steps):
            tw.step(n, 1)
            tw.observe_accumulate(0.2, n % 4 == 0)
            if (n + 1) % 4 == 0:
                kept = tw.observe_rms().copy()
        util.assert_equal_arrays(g.observe_rms_window(), kept, f"N={N}: last completed window")
        util.assert_equal_arrays(g.observe_rms(), tw.observe_rms(), f"N={N}: window being accumulated")
    else:
        g.step(0, steps)
        tw.step(0, steps)
    for id_ in po.FIELDS3D:
        util.assert_equal_arrays(g.get(id_), tw.get(id_), f"N={N} fp{prec} array {id_} after {steps} steps")
    g.close()


@pytest.mark.parametrize("seed", range(12))
def test_group_random_configurations_equal_twin(seed, pair_env):
    from tests.test_gpu_fused import _random_case
    rng = np.random.default_rng(4000 + seed)
    N, npml, prec, has_mu, rows, chunk = _random_case(rng)
    N = (N[0], N[1], max(N[2], 8))
    n_slabs = int(rng.integers(1, min(6, N[2]) + 1))
    mode = int(rng.choice([E.MODE_FULL, E.MODE_LEAN, E.MODE_FUSED]))
    K = po.consts()
    d = (5e-8, 6e-8, 7e-8)
    dt = 5e-8 / (2 * K["c0"])
    steps = 11
    kJ = int(rng.integers(0, N[2] + 1))
    g = E.EngineGroup3D(N, npml, d, dt, K["mu0"], kJ, steps, n_slabs, precision=prec, has_mu=has_mu, mode=mode)
    for m in g.members:
        m.tune(chunk, 1, rows if mode == E.MODE_FUSED else 1 + rows % 4)
    tw = po.Twin3D(prec, N, npml, d, dt, K["mu0"], kJ, steps, has_mu)
    wf = rng.standard_normal(steps)
    mag = rng.random((N[0], N[1] + 1))
    for be in (g, tw):
        for a in range(3):
            for s in range(2):
                if npml[a * 2 + s]:
                    tabs = po.cpml_face(npml[a * 2 + s], s, d[a], d[a], dt)
                    for k in range(6):
                        be.set(po.tab_id(a, s, k), tabs[k])
        be.set(po.JX_WF, wf)
        be.set(po.JX_MAG, mag)
        be.fill_materials(seed, K["eps0"], 4 * K["eps0"], K["mu0"], 3 * K["mu0"])
    g.step(0, 4)
    g.step(4, steps - 4)
    tw.step(0, steps)
    for id_ in po.FIELDS3D + po.ALL_PHI3D:
        if g.size(id_):
            util.assert_equal_arrays(g.get(id_), tw.get(id_), f"seed {seed}: group N={N} npml={npml} x{n_slabs} mode {mode} fp{prec} mu={has_mu}, array {id_}")
    g.close()


@pytest.mark.parametrize("slabs,mode", [(5, E.MODE_FUSED), (16, E.MODE_FUSED), (8, E.MODE_LEAN), (16, E.MODE_FULL)])
def test_group_many_slabs_midsize(slabs, mode, pair_env):
    """300 x 100 x 130 cells cut into up to 16 slabs (kernels long enough for the copies to overlap them),
    30 steps in two calls: same checksums as one engine on the whole grid."""
    from fdtd_amd import setup as fs
    N, steps = (300, 100, 130), 30
    K = fs.constants()
    d = (1053e-9 / 20,) * 3
    dt = d[0] / (2 * K["c0"])
    ids = (E.DX, E.DY, E.DZ, E.BX, E.BY, E.BZ)

    def prepare(whole, members):
        wf, mag = fs.default_source3d(N[0], N[1], steps, d, dt)
        for m in members:
            fs.load_cpml3d(m, d, dt)
            m.set(E.JX_WF, wf)
            m.set(E.JX_MAG, mag)
        whole.fill_materials(99, K["eps0"], 4 * K["eps0"], K["mu0"], 2 * K["mu0"])

    one = fdtd_amd.Engine3D(N, (10,) * 6, d, dt, K["mu0"], 12, steps, precision=32, has_mu=True, mode=mode)
    prepare(one, [one])
    one.step(0, steps)
    want = [one.checksum(i) for i in ids]
    one.close()
    g = E.EngineGroup3D(N, (10,) * 6, d, dt, K["mu0"], 12, steps, slabs, precision=32, has_mu=True, mode=mode)
    prepare(g, g.members)
    g.step(0, 7)
    g.step(7, steps - 7)
    assert [g.checksum(i) for i in ids] == want
    g.close()


def test_group_refusals():
    meta, _, _, _ = util.load_fixture("g1_3d_cpml")
    with pytest.raises(E.EngineError, match="cannot split"):
        group_from_meta(meta, 32, meta["N"][2] + 1, E.MODE_FUSED)
    with pytest.raises(E.EngineError, match="cannot split"):
        group_from_meta(meta, 32, 0, E.MODE_FUSED)
    with pytest.raises(E.EngineError, match="device"):
        group_from_meta(meta, 32, 2, E.MODE_FUSED, devices=[0, 4096])
    L = E.lib()
    desc = E.Desc3D(C.sizeof(E.Desc3D), 24, 20, 28, (C.c_int32 * 6)(4, 4, 4, 4, 4, 4), 6, 8, 32, 0, 2, 3, 28, -1, 5e-8, 5e-8, 5e-8, 8e-17, 4e-7 * np.pi)
    h = C.c_void_p()
    assert L.fdtd3d_group_create(C.byref(desc), 2, None, C.byref(h)) == 1 and b"whole grid" in L.fdtd_last_error()
    assert L.fdtd3d_group_step(None, 0, 1) == 1 and L.fdtd3d_group_slabs(None) == 0 and L.fdtd3d_group_destroy(None) == 0


def test_group_events_are_recorded_on_streams_of_their_own_slab(monkeypatch):
    """HIP refuses to record an event on a stream of another device.  With every slab on device 0 the
    runtime cannot notice, so the group checks the pairing itself: each record of a step passes the
    check, and a group whose arrival events are filed under the receiving slab (the layout that fails
    between real devices) is refused at its first step."""
    meta, ins, _, _ = util.load_fixture("g1_3d_cpml")
    L = E.lib()
    L.fdtd3d_group_records_checked.restype = C.c_long
    for mode in (E.MODE_FUSED, E.MODE_LEAN):
        g = group_from_meta(meta, 32, 3, mode)
        util.feed(g, ins)
        g.step(0, 2)
        g.sync()
        # per step: edge_done and top_done on each of 3 slabs, one arrival event per direction and interface
        assert L.fdtd3d_group_records_checked(g.h) == 2 * (3 + 3 + 2 + 2)
        g.close()
    L.fdtd3d_group_debug_misown_events(1)          # a test hook of the library, not part of the public header
    try:
        for mode in (E.MODE_FUSED, E.MODE_LEAN):
            g = group_from_meta(meta, 32, 3, mode)
            util.feed(g, ins)
            with pytest.raises(E.EngineError, match="recorded on a stream of slab"):
                g.step(0, 1)
            g.close()
    finally:
        L.fdtd3d_group_debug_misown_events(0)


def test_group_two_step_launches_with_observed_steps_and_resume(monkeypatch):
    """Overlapping slabs: pairs, a single step, pairs again, observed steps, a download of
    everything, an upload into a fresh group and more pairs -- every array equals the twin's throughout."""
    monkeypatch.setenv("FDTD_PAIR", "1")
    meta, ins, _, _ = util.load_fixture("g1_3d_cpml")
    g = group_from_meta(meta, 32, 3, E.MODE_FUSED)
    assert g.pairs
    tw = util.twin3d_from_meta(meta, 32)
    util.feed(g, ins)
    util.feed(tw, ins)
    g.step(0, 7)
    tw.step(0, 7)
    g.step_observed(7, 6, 0.1, 4)
    for n in range(7, 13):
        tw.step(n, 1)
        tw.observe_accumulate(0.1, n % 4 == 0)
        if (n + 1) % 4 == 0:
            kept = tw.observe_rms().copy()
    util.assert_equal_arrays(g.observe_rms_window(), kept, "window completed at step 11")
    util.assert_equal_arrays(g.observe_rms(), tw.observe_rms(), "window being accumulated after step 12")
    state = {id_: g.get(id_) for id_ in po.FIELDS3D + po.ALL_PHI3D if g.size(id_)}
    for id_, a in state.items():
        util.assert_equal_arrays(a, tw.get(id_), f"array {id_} after 13 steps")
    g.close()
    g2 = group_from_meta(meta, 32, 2, E.MODE_FUSED)
    assert g2.pairs
    util.feed(g2, ins)
    for id_, a in state.items():
        if id_ not in (E.EX, E.EY, E.EZ, E.HX, E.HY, E.HZ):
            g2.set(id_, a)
    g2.step(13, 8)
    tw.step(13, 8)
    for id_ in po.FIELDS3D + po.ALL_PHI3D:
        if g2.size(id_):
            util.assert_equal_arrays(g2.get(id_), tw.get(id_), f"array {id_} after the resumed run")
    g2.close()


@pytest.mark.parametrize("mode,pairs", [(E.MODE_FUSED, None), (E.MODE_FUSED, "1"), (E.MODE_LEAN, None), (E.MODE_FULL, None)])
def test_group_planes_travel_by_rccl(mode, pairs, monkeypatch):
    """The same choreographies with ncclSend / ncclRecv in place of the device-to-device copies.  On a one-GPU box every
    slab has the rank of device 0, so each exchange is a rank sending to itself and receiving from itself -- the RCCL
    calls, their grouping, the receiving side's streams and arrival events are the ones a multi-GPU node runs."""
    if pairs:
        monkeypatch.setenv("FDTD_PAIR", pairs)
    meta, ins, _, _ = util.load_fixture("g1_3d_cpml")
    L = E.lib()
    L.fdtd3d_group_default_transport(1)
    try:
        g = group_from_meta(meta, 32, 3, mode)
    finally:
        L.fdtd3d_group_default_transport(-1)
    assert L.fdtd3d_group_transport(g.h) == 1 and g.pairs == bool(pairs)
    tw = util.twin3d_from_meta(meta, 32)
    util.feed(g, ins)
    util.feed(tw, ins)
    g.step(0, 9)
    g.step(9, 4)
    tw.step(0, 13)
    for id_ in po.FIELDS3D + po.ALL_PHI3D:
        if g.size(id_):
            util.assert_equal_arrays(g.get(id_), tw.get(id_), f"RCCL transport, mode {mode}, pairs {pairs}: array {id_}")
    g.close()
    plain = group_from_meta(meta, 32, 2, mode)
    assert L.fdtd3d_group_transport(plain.h) == 0
    plain.close()


@pytest.mark.parametrize("name", ["g1_3d_cpml", "g2_3d_cavity"])
@pytest.mark.parametrize("prec", [32, 64])
@pytest.mark.parametrize("n_slabs,window", [(2, 4), (3, 5), (3, 1)])
def test_group_window_inside_two_step_launches_on_slabs(name, prec, n_slabs, window, monkeypatch):
    """Overlapping slabs with the RMS window advanced inside their two-step launches: a slab below another one leaves
    the second step's contribution of its top layer out and adds it once the planes from above have arrived.  Both
    windows (accumulating, last completed), the frame and every array equal the twin's, through runs of odd and even
    length that start on odd and even steps."""
    monkeypatch.setenv("FDTD_PAIR", "1")
    meta, ins, _, _ = util.load_fixture(name)
    g = group_from_meta(meta, prec, n_slabs, E.MODE_FUSED)
    assert g.pairs
    tw = util.twin3d_from_meta(meta, prec)
    util.feed(g, ins)
    util.feed(tw, ins)
    n, kept = 0, None
    for count in (1, 6, 7, 2, 8):
        g.step_observed(n, count, 0.2, window)
        for m in range(n, n + count):
            tw.step(m, 1)
            tw.observe_accumulate(0.2, m % window == 0)
            if (m + 1) % window == 0:
                kept = tw.observe_rms().copy()
        n += count
        if n % window != 0:
            util.assert_equal_arrays(g.observe_rms(), tw.observe_rms(), f"{name} x{n_slabs}: window being accumulated after step {n - 1}")
        if kept is not None:
            util.assert_equal_arrays(g.observe_rms_window(), kept, f"{name} x{n_slabs}: last completed window after step {n - 1}")
    a, b = g.observe_frame()
    ta, tb = tw.observe_frame()
    util.assert_equal_arrays(a, ta, "centred Ex")
    util.assert_equal_arrays(b, tb, "|E|")
    for id_ in po.FIELDS3D + po.ALL_PHI3D:
        if g.size(id_):
            util.assert_equal_arrays(g.get(id_), tw.get(id_), f"{name} x{n_slabs}: array {id_} after {n} observed steps")
    g.close()
