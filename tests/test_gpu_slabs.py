"""z-slab engines on one GPU: several slab engines in one process, ghost planes moved by device
copies exactly where the RCCL send/recv pairs sit in fdtd_amd.slab.SlabStepper; the assembled
result must equal the reference vectors / the whole-grid engine bit for bit."""
import numpy as np
import pytest

import fdtd_amd
from fdtd_amd import engine as E
from fdtd_amd import slab
from oracle import pyoracle as po
from tests import util

pytestmark = pytest.mark.gpu


def run_slabs(meta, ins, parts, prec, mode, steps, has_mu=False, extra_feed=None):
    engs = [fdtd_amd.Engine3D(meta["N"], meta["npml"], (meta["dx"], meta["dy"], meta["dz"]), meta["dt"], meta["mu0"],
                              meta["k_Jx"], meta["steps"], precision=prec, mode=mode, z_range=p, has_mu=has_mu) for p in parts]
    for e in engs:
        util.feed(e, ins)
        if extra_feed:
            extra_feed(e)
    pl = [{w: [E.halo_tensor(e, w, c) for c in (0, 1)] for w in range(4)} for e in engs]
    for n in range(steps):
        for e in engs:
            e.half_e(n, E.PART_INTERIOR)
        for e in engs:
            e.half_e(n, E.PART_EDGE)
        for r in range(1, len(engs)):
            for c in (0, 1):
                pl[r - 1][E.HALO_RECV_HI][c].copy_(pl[r][E.HALO_SEND_DOWN][c])
        for e in engs:
            e.half_h(E.PART_INTERIOR)
        for e in engs:
            e.half_h(E.PART_EDGE)
        for r in range(len(engs) - 1):
            for c in (0, 1):
                pl[r + 1][E.HALO_RECV_LO][c].copy_(pl[r][E.HALO_SEND_UP][c])
    return engs


def assemble(engs, id_):
    buf = np.full(engs[0].size(id_), np.nan)
    for e in engs:
        e.get(id_, out=buf)
    return buf


@pytest.mark.parametrize("mode", [E.MODE_FULL, E.MODE_LEAN])
@pytest.mark.parametrize("name,world", [("g1_3d_cpml", 2), ("g3_3d_hetero", 3), ("g2_3d_cavity", 4), ("g3_3d_hetero", 8)])
def test_slabs_reproduce_reference(name, world, mode):
    meta, ins, outs, _ = util.load_fixture(name)
    steps = meta["steps"]
    engs = run_slabs(meta, ins, slab.partition(meta["N"][2], world), 64, mode, steps)
    for (step, id_), want in outs.items():
        if step == steps:
            util.assert_equal_arrays(assemble(engs, id_), want, f"{name} x{world} slabs, array {id_}")


def test_slabs_fp32_mu_grid_equal_whole_grid():
    meta, ins, outs, _ = util.load_fixture("g3_3d_hetero")
    K = meta
    fill = lambda e: e.fill_materials(5, K["eps0"], 3 * K["eps0"], K["mu0"], 2 * K["mu0"])
    whole = run_slabs(meta, ins, [(0, meta["N"][2])], 32, E.MODE_LEAN, 30, has_mu=True, extra_feed=fill)
    parts = run_slabs(meta, ins, [(0, 5), (5, 6), (6, 19), (19, 26)], 32, E.MODE_LEAN, 30, has_mu=True, extra_feed=fill)
    for id_ in po.FIELDS3D + po.ALL_PHI3D:
        util.assert_equal_arrays(assemble(parts, id_), assemble(whole, id_), f"uneven slabs, array {id_}")
    # order-independent checksums add up across slabs
    for id_ in (po.DX, po.DZ, po.BY):
        assert sum(e.checksum(id_) for e in parts) % 2 ** 64 == whole[0].checksum(id_)
