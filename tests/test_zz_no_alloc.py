"""fdtd3d_reduce_static: sums of squares and checksums of several field arrays from one call that
allocates nothing (its partial sums live in a file-scope device array of the library).  It must
return exactly what fdtd3d_sumsq / fdtd3d_checksum return, also when the shared array still holds
the partial sums of an earlier, larger call."""
import ctypes as C

import numpy as np
import pytest

import fdtd_amd
from fdtd_amd import engine as E
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _engine(N, npml, prec, mode, steps, seed):
    K = po.consts()
    d = (5e-8, 6e-8, 7e-8)
    dt = 5e-8 / (2 * K["c0"])
    eng = fdtd_amd.Engine3D(N, npml, d, dt, K["mu0"], N[2] // 2, steps, precision=prec, has_mu=True, mode=mode)
    rng = np.random.default_rng(seed)
    for a in range(3):
        for s in range(2):
            if npml[a * 2 + s]:
                tabs = po.cpml_face(npml[a * 2 + s], s, d[a], d[a], dt)
                for k in range(6):
                    eng.set(po.tab_id(a, s, k), tabs[k])
    eng.set(po.JX_WF, rng.standard_normal(steps))
    eng.set(po.JX_MAG, rng.random((N[0], N[1] + 1)))
    eng.fill_materials(seed, K["eps0"], 4 * K["eps0"], K["mu0"], 3 * K["mu0"])
    eng.step(0, steps)
    return eng


def _static(eng, ids):
    L = E.lib()
    n = len(ids)
    arr = (C.c_int * max(n, 1))(*ids)
    sums = (C.c_double * max(n, 1))()
    cks = (C.c_uint64 * max(n, 1))()
    rc = L.fdtd3d_reduce_static(eng.h, arr, C.c_int(n), sums, cks, None)
    return rc, list(sums)[:n], list(cks)[:n]


def test_reduce_static_equals_allocating_entry_points_twice_in_a_row():
    big = _engine((72, 40, 44), (4, 3, 5, 2, 3, 4), 32, E.MODE_FUSED, 9, 11)
    ids_big = [E.DX, E.DY, E.DZ, E.BX, E.BY, E.BZ, E.EPS_X, E.MU_Z]
    rc, sums, cks = _static(big, ids_big)
    assert rc == 0, E.lib().fdtd_last_error()
    assert sums == [big.sumsq(i) for i in ids_big]           # bit for bit: same partial sums, same order
    assert cks == [big.checksum(i) for i in ids_big]
    assert all(v > 0 for v in sums[:6])
    # second call, smaller input (fewer, smaller arrays, other precision and storage mode): the shared
    # array still holds the first call's partial sums and checksum words
    small = _engine((24, 20, 28), (0, 2, 3, 0, 2, 2), 64, E.MODE_FULL, 7, 12)
    ids_small = [E.EZ, E.HY]
    rc, sums, cks = _static(small, ids_small)
    assert rc == 0, E.lib().fdtd_last_error()
    assert sums == [small.sumsq(i) for i in ids_small]
    assert cks == [small.checksum(i) for i in ids_small]
    # and the first engine again, only one output requested
    L = E.lib()
    one = (C.c_uint64 * 1)()
    assert L.fdtd3d_reduce_static(big.h, (C.c_int * 1)(E.BY), 1, None, one, None) == 0
    assert one[0] == big.checksum(E.BY)


def test_reduce_static_refusals():
    eng = _engine((24, 20, 28), (2,) * 6, 32, E.MODE_LEAN, 2, 5)
    L = E.lib()
    rc, _, _ = _static(eng, [E.DX] * 17)
    assert rc == 1 and b"at most 16" in L.fdtd_last_error()
    rc, _, _ = _static(eng, [E.EX])                # E is not resident in LEAN storage
    assert rc == 1 and b"not a resident field" in L.fdtd_last_error()
    rc, _, _ = _static(eng, [E.DX, 99])
    assert rc == 1
    assert L.fdtd3d_reduce_static(None, None, 0, None, None, None) == 1
    rc, sums, cks = _static(eng, [])
    assert rc == 0 and sums == [] and cks == []
