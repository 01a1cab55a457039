"""Checkpoint and resume through the array exchange alone: the state downloaded after n steps (fields and
every CPML auxiliary; in 2-D also the Lorentz histories), uploaded into a fresh engine -- of the same or of
another storage mode, whole or cut into slabs -- and stepped on, must give the bits of the uninterrupted
run.  The reference keeps its state in private vectors and cannot be stopped (fdtd3d/include/FDTD3D.h:31-108);
this is the round-trip property of the boundary that replaces them."""
import numpy as np
import pytest

import fdtd_amd
from fdtd_amd import engine as E
from oracle import pyoracle as po
from tests import util

pytestmark = pytest.mark.gpu


def _inputs3d(meta, be, ins):
    util.feed(be, ins)


@pytest.mark.parametrize("first,second", [(E.MODE_FULL, E.MODE_FULL), (E.MODE_FUSED, E.MODE_FUSED), (E.MODE_FULL, E.MODE_FUSED),
                                          (E.MODE_FUSED, E.MODE_LEAN), (E.MODE_LEAN, E.MODE_FULL)])
@pytest.mark.parametrize("prec", [32, 64])
def test_resume_3d_equals_uninterrupted_run(first, second, prec):
    meta, ins, _, _ = util.load_fixture("g3_3d_hetero")
    args = (meta["N"], meta["npml"], (meta["dx"], meta["dy"], meta["dz"]), meta["dt"], meta["mu0"], meta["k_Jx"], meta["steps"])
    n1, n2 = 9, 21                                   # an odd number of steps: the fused engine stops on its second buffer set
    straight = fdtd_amd.Engine3D(*args, precision=prec, mode=first)
    util.feed(straight, ins)
    straight.step(0, n2)
    a = fdtd_amd.Engine3D(*args, precision=prec, mode=first)
    util.feed(a, ins)
    a.step(0, n1)
    state_ids = [i for i in (po.FIELDS3D + po.ALL_PHI3D) if a.size(i)]
    saved = {i: a.get(i) for i in state_ids}
    a.close()
    b = fdtd_amd.Engine3D(*args, precision=prec, mode=second)
    util.feed(b, ins)
    for i, v in saved.items():
        if second != E.MODE_FULL and i < 12 and (i // 3) in (1, 3):
            continue                                 # E and H are derived in lean storage
        b.set(i, v)
    b.step(n1, n2 - n1)
    for i in state_ids:
        util.assert_equal_arrays(b.get(i), straight.get(i), f"resumed {first}->{second} fp{prec}, array {i}")


def test_resume_3d_into_a_slab_group():
    meta, ins, _, _ = util.load_fixture("g1_3d_cpml")
    args = (meta["N"], meta["npml"], (meta["dx"], meta["dy"], meta["dz"]), meta["dt"], meta["mu0"], meta["k_Jx"], meta["steps"])
    n1, n2 = 7, 18
    straight = fdtd_amd.Engine3D(*args, precision=32, mode=E.MODE_FUSED)
    util.feed(straight, ins)
    straight.step(0, n2)
    a = fdtd_amd.Engine3D(*args, precision=32, mode=E.MODE_FUSED)
    util.feed(a, ins)
    a.step(0, n1)
    ids = [i for i in (po.FIELDS3D + po.ALL_PHI3D) if a.size(i)]
    saved = {i: a.get(i) for i in ids}
    g = E.EngineGroup3D(*args, 3, precision=32, mode=E.MODE_FUSED)
    util.feed(g, ins)
    for i, v in saved.items():
        if i < 12 and (i // 3) in (1, 3):
            continue
        g.set(i, v)
    g.step(n1, n2 - n1)
    for i in ids:
        util.assert_equal_arrays(g.get(i), straight.get(i), f"resumed into 3 slabs, array {i}")


@pytest.mark.parametrize("first,second", [(E.MODE2D_FULL, E.MODE2D_FULL), (E.MODE2D_FUSED, E.MODE2D_FUSED), (E.MODE2D_FULL, E.MODE2D_FUSED),
                                          (E.MODE2D_FUSED, E.MODE2D_FULL)])
def test_resume_2d_equals_uninterrupted_run(first, second):
    meta, ins, _, _ = util.load_fixture("g4_2d_slab")
    args = (meta["Nx"], meta["Nz"], meta["npml"], meta["dx"], meta["dz"], meta["dt"], meta["mu0"], meta["eps0"], meta["i_Jy"], meta["k_Jy"], meta["steps"])
    n1, n2 = 33, 80
    straight = fdtd_amd.Engine2D(*args, precision=32, mode=first)
    util.feed(straight, ins)
    straight.step(0, n2)
    a = fdtd_amd.Engine2D(*args, precision=32, mode=first)
    util.feed(a, ins)
    a.step(0, n1)
    ids = [i for i in (po.FIELDS2D + po.ALL_PHI2D) if a.size(i)]
    saved = {i: a.get(i) for i in ids}
    b = fdtd_amd.Engine2D(*args, precision=32, mode=second)
    util.feed(b, ins)
    for i, v in saved.items():
        b.set(i, v)
    b.step(n1, n2 - n1)
    for i in (po.D2_DY, po.D2_EY, po.D2_BX, po.D2_BZ, po.D2_HX, po.D2_HZ) + tuple(po.ALL_PHI2D):
        if b.size(i):
            util.assert_equal_arrays(b.get(i), straight.get(i), f"2-D resumed {first}->{second}, array {i}")
