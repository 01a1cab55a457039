"""Error behaviour of the C ABI: the reference's solvers have no error path (private void members,
fdtd3d/include/FDTD3D.h:190-195, fdtd2d/include/FDTD2D.h:88-89), so every refusal here is one the
boundary adds; each must come back as a status code with a message, never as a crash."""
import ctypes as C

import numpy as np
import pytest

import fdtd_amd
from fdtd_amd import engine as E

pytestmark = pytest.mark.gpu

MU0 = 4e-7 * np.pi
GOOD3 = dict(N=(24, 20, 28), npml=(4,) * 6, d=(5e-8,) * 3, dt=8e-17, mu0=MU0, k_Jx=6, n_wf=8)


def make3(**over):
    kw = dict(GOOD3, **over)
    return fdtd_amd.Engine3D(kw["N"], kw["npml"], kw["d"], kw["dt"], kw["mu0"], kw["k_Jx"], kw["n_wf"],
                             precision=kw.get("precision", 32), has_mu=kw.get("has_mu", False),
                             mode=kw.get("mode", E.MODE_FULL), z_range=kw.get("z_range"), device=kw.get("device", -1))


@pytest.mark.parametrize("over, word", [
    (dict(precision=16), "precision"),
    (dict(N=(1, 20, 28)), "too small"),
    (dict(npml=(12, 12, 4, 4, 4, 4)), "do not fit"),
    (dict(npml=(4, 4, 4, 4, 4, 65)), "CPML layer counts"),
    (dict(z_range=(10, 10)), "bad slab"),
    (dict(z_range=(0, 29)), "bad slab"),
    (dict(k_Jx=29), "source plane"),
    (dict(mode=7), "storage mode"),
    (dict(dt=0.0), "positive"),
    (dict(device=4096), "device"),
])
def test_bad_descriptors_are_refused_3d(over, word):
    with pytest.raises(E.EngineError, match=word):
        make3(**over)


def test_descriptor_size_is_checked():
    L = E.lib()
    desc = E.Desc3D(C.sizeof(E.Desc3D) - 8, 24, 20, 28, (C.c_int32 * 6)(4, 4, 4, 4, 4, 4), 6, 8, 32, 0, 0, 0, 28, -1, 5e-8, 5e-8, 5e-8, 8e-17, MU0)
    h = C.c_void_p()
    assert L.fdtd3d_create(C.byref(desc), C.byref(h)) == 1 and not h.value
    assert b"descriptor size" in L.fdtd_last_error()
    d2 = E.Desc2D(4, 32, 32, (C.c_int32 * 4)(4, 4, 4, 4), 16, 16, 8, 32, -1, 5e-8, 5e-8, 8e-17, MU0, 8.854e-12, 0, 0)
    assert L.fdtd2d_create(C.byref(d2), C.byref(h)) == 1 and not h.value


def test_null_handles_and_null_outputs():
    L = E.lib()
    null = C.c_void_p(0)
    assert L.fdtd3d_step(null, 0, 1, None) == 1
    assert L.fdtd3d_half_e(null, 0, 0, None) == 1
    assert L.fdtd2d_step(null, 0, 1, None) == 1
    assert L.fdtd3d_array_size(null, 0) == 0
    assert L.fdtd3d_destroy(null) == 0 and L.fdtd2d_destroy(null) == 0
    assert L.fdtd3d_create(None, C.byref(null)) == 1
    assert b"null" in L.fdtd_last_error()


def test_array_exchange_refusals_3d():
    eng = make3(mode=E.MODE_LEAN)
    with pytest.raises(E.EngineError, match="expected"):
        eng.set(E.DX, np.zeros(5))
    assert E.lib().fdtd3d_set_array(eng.h, 999, None, C.c_size_t(0)) == 1
    assert b"unknown array id" in E.lib().fdtd_last_error()
    with pytest.raises(E.EngineError, match="not resident"):
        eng.set(E.EX, np.zeros(eng.size(E.EX)))
    assert eng.size(E.MU_X) == 0                       # no mu grid on this engine
    with pytest.raises(E.EngineError):
        eng.fused(0)                                   # not a FUSED engine
    with pytest.raises(E.EngineError, match="bad part"):
        eng.half_e(0, part=9)


def test_bad_descriptors_are_refused_2d():
    def make(**over):
        kw = dict(Nx=40, Nz=48, npml=(4, 4, 4, 4), dx=5e-8, dz=5e-8, dt=8e-17, mu0=MU0, eps0=8.854e-12, i_Jy=20, k_Jy=10, n_wf=8)
        kw.update(over)
        return fdtd_amd.Engine2D(kw["Nx"], kw["Nz"], kw["npml"], kw["dx"], kw["dz"], kw["dt"], kw["mu0"], kw["eps0"],
                                 kw["i_Jy"], kw["k_Jy"], kw["n_wf"], precision=kw.get("precision", 32), mode=kw.get("mode", 0),
                                 device=kw.get("device", -1))
    make().close()
    for over, word in [(dict(precision=8), "precision"), (dict(mode=3), "mode"), (dict(Nx=1), "too small"),
                       (dict(npml=(20, 20, 4, 4)), "do not fit"), (dict(i_Jy=40), "wall"), (dict(eps0=0.0), "positive"),
                       (dict(device=4096), "device")]:
        with pytest.raises(E.EngineError, match=word):
            make(**over)


def test_calls_leave_the_callers_device_current():
    import torch
    before = torch.cuda.current_device()
    eng = make3(device=0, mode=E.MODE_FUSED)
    eng.step(0, 2)
    eng.get(E.DX)
    assert torch.cuda.current_device() == before
    eng.close()


@pytest.mark.parametrize("mode", [E.MODE_FULL, E.MODE_LEAN, E.MODE_FUSED])
def test_empty_waveform_and_steps_past_its_end(mode):
    """No waveform at all: nothing is ever injected and the zero state stays zero.  A waveform shorter than
    the run: steps past its end inject nothing, exactly like the twin (the reference indexes its waveform
    with the step number, fdtd3d/src/FDTD3D.cpp:482, and never runs past it)."""
    from oracle import pyoracle as po
    quiet = make3(n_wf=0, mode=mode)
    assert quiet.size(E.JX_WF) == 0
    quiet.set(E.JX_WF, np.zeros(0))
    quiet.set(E.JX_MAG, np.ones((24, 21)))
    quiet.step(0, 5)
    assert quiet.sumsq(E.DX) == 0.0 and quiet.sumsq(E.BY) == 0.0
    n_wf, steps = 4, 9
    eng = make3(n_wf=n_wf, mode=mode)
    tw = po.Twin3D(32, GOOD3["N"], GOOD3["npml"], GOOD3["d"], GOOD3["dt"], GOOD3["mu0"], GOOD3["k_Jx"], n_wf, False)
    rng = np.random.default_rng(8)
    wf, mag = rng.standard_normal(n_wf), rng.random((24, 21))
    for be in (eng, tw):
        for a in range(3):
            for s_ in range(2):
                tabs = po.cpml_face(4, s_, GOOD3["d"][a], GOOD3["d"][a], GOOD3["dt"])
                for k in range(6):
                    be.set(po.tab_id(a, s_, k), tabs[k])
        be.set(E.JX_WF, wf)
        be.set(E.JX_MAG, mag)
        be.fill_materials(3, 8.854e-12, 3 * 8.854e-12, MU0, MU0)
    eng.step(0, steps)
    tw.step(0, steps)
    for id_ in (E.DX, E.DY, E.DZ, E.BX, E.BY, E.BZ):
        a, b = eng.get(id_), tw.get(id_)
        assert np.array_equal(a, b) and (id_ != E.DX or np.abs(a).max() > 0)


def test_timeline_launch_shape_fill_and_placement_calls(monkeypatch):
    """The measurement hooks added beside the step functions: per-launch timeline, launch shape, array fill, and the
    placement queries -- what they report, and what they refuse."""
    L = E.lib()
    null = C.c_void_p(0)
    n = C.c_int(-1)
    assert L.fdtd3d_kernel_timeline(null, 0, None, None, None, None, C.byref(n)) == 1
    assert L.fdtd3d_launch_shape(null, (C.c_int * 6)()) == 1 and L.fdtd3d_fill_array(null, 0, 1, C.c_double(0), C.c_double(1)) == 1
    assert L.fdtd3d_grant_peer_access(null, 0) == 1
    L.fdtd3d_spread_gap_gib.restype = C.c_double
    assert L.fdtd3d_spread_gap_gib(null) == 0.0
    eng = make3(mode=E.MODE_FUSED, z_range=(4, 20))
    assert eng.kernel_timeline() == []                               # nothing timed yet
    eng.kernel_timing(True)
    for step in range(3):
        eng.fused(step, E.PART_EDGE)
        eng.fused(step, E.PART_TOP)
        eng.fused_top()
        eng.fused(step, E.PART_INTERIOR)
    tl = eng.kernel_timeline()
    assert [(k, p) for k, p, _, _ in tl] == [(2, E.PART_EDGE), (2, E.PART_TOP), (1, E.PART_EDGE), (2, E.PART_INTERIOR)] * 3
    starts = [t for _, _, t, _ in tl]
    assert starts == sorted(starts) and starts[0] == 0.0 and all(d > 0 for _, _, _, d in tl)
    assert len(eng.kernel_timeline()) == 12                          # reading the timeline leaves the records in place
    assert eng.kernel_time(2)[0] == 9 and eng.kernel_time(1)[0] == 3 and eng.kernel_timeline() == []
    shape = eng.launch_shape()
    assert shape["rows_per_group"] in (8, 16) and shape["edge_planes"] == 1 and shape["row_pitch"] % 256 == 0 and shape["planes_per_group"] >= 1
    eng.tune(3, 0, 5)
    assert (eng.launch_shape()["planes_per_group"], eng.launch_shape()["xcd_run"], eng.launch_shape()["rows_per_group"]) == (3, 0, 5)
    # fill_array: resident field arrays only
    eng.fill_array(E.DX, 3, -1.0, 1.0)
    assert eng.sumsq(E.DX) > 0
    with pytest.raises(E.EngineError, match="not a resident field"):
        eng.fill_array(E.EX, 3, 0.0, 1.0)                            # E is not stored in FUSED mode
    with pytest.raises(E.EngineError, match="not a resident field"):
        eng.fill_array(99, 3, 0.0, 1.0)
    # a small engine is one plain allocation; granting access to its own device is a no-op either way
    assert eng.spread_gap_gib() == 0.0 and L.fdtd3d_grant_peer_access(eng.h, 0) == 0
    # the same engine backed through the virtual-memory API steps to the same bits
    want = [eng.checksum(i) for i in (E.DX, E.DY, E.BZ)]
    monkeypatch.setenv("FDTD_SPREAD", "1")
    vm = make3(mode=E.MODE_FUSED, z_range=(4, 20))
    vm.tune(3, 0, 5)
    vm.fill_array(E.DX, 3, -1.0, 1.0)
    for e in (eng, vm):
        e.fused(3, E.PART_ALL)
        e.fused_top()
    want = [eng.checksum(i) for i in (E.DX, E.DY, E.BZ)]
    assert [vm.checksum(i) for i in (E.DX, E.DY, E.BZ)] == want and L.fdtd3d_grant_peer_access(vm.h, 0) == 0
