"""Host-side pieces of the product on CPU: setup generators, library symbols, argument checks."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import fdtd_amd
from fdtd_amd import setup as fs
from oracle import pyoracle as po
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = fdtd_amd.lib()
    names = set()
    for hdr in ("fdtd_engine.h", "fdtd_setup.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(fdtd[23]?d?_[a-z0-9_]+)\s*\(", text))
    assert len(names) >= 30
    for n in sorted(names):
        assert hasattr(L, n), f"{n} declared in include/ but not exported"


def test_setup_generators_match_reference_inputs():
    """Product generators == the tables the compiled reference built (golden inputs)."""
    meta, ins, _, _ = util.load_fixture("g3_3d_hetero")
    d = (meta["dx"], meta["dy"], meta["dz"])
    for a in range(3):
        for s in range(2):
            t = fs.cpml_tables(meta["npml"][a * 2 + s], s, d[a], d[a], meta["dt"])
            for k in range(6):
                util.assert_equal_arrays(t[k], ins[po.tab_id(a, s, k)], f"table {a} {s} {k}")
    wf, mag = fs.default_source3d(meta["N"][0], meta["N"][1], meta["steps"], d, meta["dt"])
    util.assert_equal_arrays(wf, ins[po.JX_WF], "wf")
    util.assert_equal_arrays(mag, ins[po.JX_MAG], "mag")
    m2, i2, _, _ = util.load_fixture("g4_2d_slab")
    dd = (m2["dx"], m2["dz"])
    for a in range(2):
        for s in range(2):
            t = fs.cpml_tables(m2["npml"][a * 2 + s], s, dd[a], m2["dz"], m2["dt"])
            for k in range(6):
                util.assert_equal_arrays(t[k], i2[po.D2_TAB + (a * 2 + s) * 6 + k], f"2d table {a} {s} {k}")
    util.assert_equal_arrays(fs.default_source2d(m2["steps"], m2["dx"], m2["dz"], m2["dt"]), i2[po.D2_JY], "Jy")
    for q, c in enumerate(fs.default_lorentz2d(m2["Nx"], m2["Nz"], m2["dt"])):
        util.assert_equal_arrays(c, i2[po.D2_C1 + q], f"C{q + 1}")
    K, Kt = fs.constants(), po.consts()
    assert K == Kt


def test_default_materials_match_twin():
    """Air-sphere material generator: product == twin on a grid the sphere surface crosses
    (the twin itself is pinned to the compiled reference in test_oracle_ref.py)."""
    N, d = (100, 14, 12), (5.265e-8,) * 3
    for a, b in zip(fs.default_materials3d(N, d), po.default_materials3d(N, d)):
        util.assert_equal_arrays(a, b, "eps")
        assert len(np.unique(a)) > 2


def test_create_fails_loudly_without_device_or_with_bad_arguments():
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    K = fs.constants()
    with pytest.raises(fdtd_amd.EngineError):
        fdtd_amd.Engine3D((8, 8, 8), (4, 4, 0, 0, 0, 0), (1e-7,) * 3, 1e-16, K["mu0"], 4, 3)      # layers do not fit
    with pytest.raises(fdtd_amd.EngineError):
        fdtd_amd.Engine2D(40, 40, (0,) * 4, 1e-7, 1e-7, 1e-16, K["mu0"], K["eps0"], 40, 3, 3)      # source on the wall column
    if not has_gpu:
        with pytest.raises(fdtd_amd.EngineError, match="device|HIP|hip"):
            fdtd_amd.Engine3D((8, 8, 8), (0,) * 6, (1e-7,) * 3, 1e-16, K["mu0"], 4, 3)


def test_group_and_device_count_refuse_without_touching_a_device():
    """Argument checks of the slab-group entry points come before any device call, so they answer on a
    box without a GPU too; fdtd_device_count() reports what is visible (0 here on the CPU box)."""
    import ctypes as C
    from fdtd_amd import engine as E
    L = fdtd_amd.lib()
    h = C.c_void_p()
    assert L.fdtd3d_group_create(None, 2, None, C.byref(h)) == 1 and not h.value
    desc = E.Desc3D(C.sizeof(E.Desc3D), 16, 16, 16, (C.c_int32 * 6)(0, 0, 0, 0, 0, 0), 4, 3, 32, 0, 2, 0, 16, -1, 1e-7, 1e-7, 1e-7, 1e-16, 1.2566e-6)
    assert L.fdtd3d_group_create(C.byref(desc), 0, None, C.byref(h)) == 1 and b"cannot split" in L.fdtd_last_error()
    assert L.fdtd3d_group_create(C.byref(desc), 17, None, C.byref(h)) == 1
    assert L.fdtd3d_group_step(None, 0, 1) == 1 and L.fdtd3d_group_sync(None) == 1
    assert L.fdtd3d_group_slabs(None) == 0 and L.fdtd3d_group_destroy(None) == 0
    n = L.fdtd_device_count()
    try:
        import torch
        assert n == (torch.cuda.device_count() if torch.cuda.is_available() else 0)
    except ImportError:
        assert n >= 0


def test_pair_scratch_bound_is_checked_on_the_host():
    """The two-step kernel forms a wave's offset into its CPML scratch memory in 32 bits: 24 row segments of a
    wave's columns (128 in fp32, 64 in fp64) per wave.  The headline launch (1024^3, 128 planes per workgroup)
    fits; the same grid cut into 8-plane chunks would need 4.5e9 elements and must be refused, not wrapped."""
    L = fdtd_amd.lib()
    L.fdtd3d_pair_scratch_fits.argtypes = [C.c_longlong, C.c_int, C.c_int]
    per_group32 = 16 * 24 * 128
    assert L.fdtd3d_pair_scratch_fits(9 * 79 * 10, 16, 32) == 1                 # 1024^3, default shape: 0.35e9 elements
    assert L.fdtd3d_pair_scratch_fits(92_000, 16, 32) == 0                      # 1024^3 in 8-plane chunks: 4.5e9
    edge = (2 ** 32 - 1) // per_group32
    assert L.fdtd3d_pair_scratch_fits(edge, 16, 32) == 1 and L.fdtd3d_pair_scratch_fits(edge + 1, 16, 32) == 0
    assert L.fdtd3d_pair_scratch_fits(2 * edge, 16, 64) == 1 and L.fdtd3d_pair_scratch_fits(2 * edge + 2, 16, 64) == 0
    assert L.fdtd3d_pair_scratch_fits(10, 16, 16) == 0 and L.fdtd3d_pair_scratch_fits(-1, 16, 32) == 0
