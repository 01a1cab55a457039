"""Properties at sizes no CPU oracle reaches (the 512^3 cavity and the 1024-wide material/CPML workloads, reduced step counts).

Bit-exactness is carried by order-independent checksums (value bits mixed with the cell index):
lean storage == full storage, slabs == whole grid, launch shape (planes per workgroup, tile order)
never changes a bit; plus the reference's own sanity heuristics (all finite, no blow-up)."""
import numpy as np
import pytest

import fdtd_amd
from fdtd_amd import engine as E
from fdtd_amd import setup as fs
from fdtd_amd import slab

pytestmark = pytest.mark.gpu
FIELDS = (E.DX, E.DY, E.DZ, E.BX, E.BY, E.BZ)


def make(N, npml, mode, has_mu, z_range=None, steps=64, dipole=False):
    K = fs.constants()
    d = (1053e-9 / 20,) * 3
    dt = d[0] / (2 * K["c0"])
    eng = fdtd_amd.Engine3D(N, npml, d, dt, K["mu0"], N[2] // 2 if dipole else 20, steps, precision=32, has_mu=has_mu, mode=mode, z_range=z_range)
    fs.load_cpml3d(eng, d, dt)
    wf, mag = fs.default_source3d(N[0], N[1], steps, d, dt)
    if dipole:
        mag[:] = 0
        mag[N[0] // 2, N[1] // 2] = 1.0
    eng.set(E.JX_WF, wf)
    eng.set(E.JX_MAG, mag)
    if has_mu:
        eng.fill_materials(1234, K["eps0"], 4 * K["eps0"], K["mu0"], 2 * K["mu0"])
    else:
        eng.fill_materials(1234, K["eps0"], K["eps0"], K["mu0"], K["mu0"])     # vacuum
    return eng


def test_vacuum_cavity_512_all_modes_equal_and_stay_bounded():
    """512^3 vacuum cavity (no CPML), dipole source: the default fused kernel, lean and full storage leave the
    same bits in D and B (an odd step count, so that the fused engine is read from its second buffer set)."""
    N, steps = (512, 512, 512), 25
    sums = {}
    for mode in (E.MODE_FULL, E.MODE_LEAN, E.MODE_FUSED):
        eng = make(N, (0,) * 6, mode, False, steps=steps, dipole=True)
        eng.step(0, steps)
        sums[mode] = [eng.checksum(i) for i in FIELDS]
        energy = [eng.sumsq(i) for i in FIELDS]
        assert all(np.isfinite(energy)) and energy[0] > 0
        eng.close()
    assert sums[E.MODE_FULL] == sums[E.MODE_LEAN] == sums[E.MODE_FUSED]


def test_hetero_cpml_slabs_equal_whole_grid_384():
    """The material-grid workload (eps/mu grids, 20-layer CPML) at 384x384x512: 4 slabs == whole grid,
    and the launch shape does not change a bit."""
    N, steps = (384, 384, 512), 12
    whole = make(N, (20,) * 6, E.MODE_LEAN, True, steps=steps)
    whole.step(0, steps // 2)
    whole.tune(3, 0)
    whole.step(steps // 2, steps - steps // 2)
    want = [whole.checksum(i) for i in FIELDS]
    e_whole = whole.sumsq(E.DX)
    whole.close()
    parts = slab.partition(N[2], 4)
    engs = [make(N, (20,) * 6, E.MODE_LEAN, True, z_range=p, steps=steps) for p in parts]
    pl = [{w: [E.halo_tensor(e, w, c) for c in (0, 1)] for w in range(4)} for e in engs]
    for n in range(steps):
        for e in engs:
            e.half_e(n, E.PART_INTERIOR)
            e.half_e(n, E.PART_EDGE)
        for r in range(1, 4):
            for c in (0, 1):
                pl[r - 1][E.HALO_RECV_HI][c].copy_(pl[r][E.HALO_SEND_DOWN][c])
        for e in engs:
            e.half_h(E.PART_INTERIOR)
            e.half_h(E.PART_EDGE)
        for r in range(3):
            for c in (0, 1):
                pl[r + 1][E.HALO_RECV_LO][c].copy_(pl[r][E.HALO_SEND_UP][c])
    got = [sum(e.checksum(i) for e in engs) % 2 ** 64 for i in FIELDS]
    assert got == want
    assert abs(sum(e.sumsq(E.DX) for e in engs) / e_whole - 1) < 1e-12


def test_energy_growth_heuristic_1024_plane_source():
    """The reference's verify_vti heuristic (energy growth ratio between frames within [0.1, 100],
    nothing non-finite) on a 1024 x 1024 x 256 slab of the headline workload, taken after the
    source's 200-step ramp so that the ratio measures propagation, not the ramp."""
    N, steps = (1024, 1024, 256), 260
    eng = make(N, (20,) * 6, E.MODE_LEAN, True, steps=steps)
    eng.step(0, 220)
    e1 = eng.sumsq(E.DX)
    eng.step(220, 40)
    e2 = eng.sumsq(E.DX)
    assert np.isfinite(e1) and np.isfinite(e2) and e1 > 0
    assert 0.1 < e2 / e1 < 100


def test_fused_equals_lean_1024_headline_workload():
    """The benchmarked configuration itself (1024^3, eps/mu grids, 20-layer CPML, fp32): the fused
    whole-step kernel and the two half-step kernels must leave identical bits in D and B after an odd
    number of steps (so that the comparison reads the second buffer set), for two launch shapes."""
    N, steps = (1024, 1024, 1024), 7
    lean = make(N, (20,) * 6, E.MODE_LEAN, True, steps=steps)
    lean.step(0, steps)
    want = [lean.checksum(i) for i in FIELDS]
    e_lean = lean.sumsq(E.DX)
    lean.close()
    assert e_lean > 0 and np.isfinite(e_lean)
    fused = make(N, (20,) * 6, E.MODE_FUSED, True, steps=steps)
    # an engine this large spreads its field arrays over physical memory: one gap in front of every large array but
    # the first (D, B twice each, eps, mu: 18 arrays), the smallest of them reported as the engine's gap
    gaps = fused.spread_gaps()
    assert len(gaps) == 17 and min(gaps) == fused.spread_gap_gib() and fused.spread_gap_gib() > 0
    fused.step(0, 4)
    fused.tune(5, 0, 8)
    fused.step(4, steps - 4)
    assert [fused.checksum(i) for i in FIELDS] == want
    fused.close()


def test_plain_allocation_when_spreading_is_off(monkeypatch):
    """FDTD_SPREAD=0: one plain allocation, same bits as the spread engine."""
    want = None
    for spread in ("0", None):
        if spread is None:
            monkeypatch.delenv("FDTD_SPREAD")
        else:
            monkeypatch.setenv("FDTD_SPREAD", spread)
        eng = make((1024, 1024, 256), (20,) * 6, E.MODE_FUSED, True, steps=4)      # 21 GB
        assert (eng.spread_gap_gib() == 0 and eng.spread_gaps() == []) if spread == "0" else eng.spread_gap_gib() > 0
        eng.step(0, 3)
        got = [eng.checksum(i) for i in FIELDS]
        assert np.isfinite(eng.sumsq(E.DX)) and (want is None or got == want)
        want = got
        eng.close()


def test_fused_slabs_equal_whole_grid_384():
    """Fused stepping on 4 slabs (lowest chunk first, top layer finished after the exchange) == the
    half-step kernels on the whole grid, eps/mu grids and 20-layer CPML."""
    N, steps = (384, 384, 512), 11
    whole = make(N, (20,) * 6, E.MODE_LEAN, True, steps=steps)
    whole.step(0, steps)
    want = [whole.checksum(i) for i in FIELDS]
    whole.close()
    parts = slab.partition(N[2], 4)
    engs = [make(N, (20,) * 6, E.MODE_FUSED, True, z_range=p, steps=steps) for p in parts]
    pl = [[{w: [E.halo_tensor(e, w, c, par) for c in (0, 1)] for w in range(4)} for par in (0, 1)] for e in engs]
    for n in range(steps):
        for e in engs:
            e.fused(n, E.PART_EDGE)
        cur = [pl[r][engs[r].parity()] for r in range(4)]
        for r in range(1, 4):
            for c in (0, 1):
                cur[r - 1][E.HALO_RECV_HI][c].copy_(cur[r][E.HALO_SEND_DOWN][c])
        for e in engs:
            e.fused(n, E.PART_TOP)
        for e in engs:
            e.fused_top()
            e.fused(n, E.PART_INTERIOR)
        for r in range(3):
            for c in (0, 1):
                cur[r + 1][E.HALO_RECV_LO][c].copy_(cur[r][E.HALO_SEND_UP][c])
    got = [sum(e.checksum(i) for e in engs) % 2 ** 64 for i in FIELDS]
    assert got == want


@pytest.mark.parametrize("mode", [E.MODE_FUSED, E.MODE_LEAN])
def test_slab_group_equals_whole_grid_512(mode):
    """A slab group (one host thread, streams, device-to-device ghost copies running beside the kernels)
    of 3 uneven slabs == one engine on the whole 512 x 512 x 640 grid, eps/mu grids and 20-layer CPML."""
    N, steps = (512, 512, 640), 15
    whole = make(N, (20,) * 6, E.MODE_LEAN, True, steps=steps)
    whole.step(0, steps)
    want = [whole.checksum(i) for i in FIELDS]
    whole.close()
    K = fs.constants()
    d = (1053e-9 / 20,) * 3
    dt = d[0] / (2 * K["c0"])
    g = E.EngineGroup3D(N, (20,) * 6, d, dt, K["mu0"], 20, steps, 3, precision=32, has_mu=True, mode=mode)
    wf, mag = fs.default_source3d(N[0], N[1], steps, d, dt)
    for m in g.members:
        fs.load_cpml3d(m, d, dt)
        m.set(E.JX_WF, wf)
        m.set(E.JX_MAG, mag)
    g.fill_materials(1234, K["eps0"], 4 * K["eps0"], K["mu0"], 2 * K["mu0"])
    g.step(0, steps)
    assert [g.checksum(i) for i in FIELDS] == want
    g.close()


def test_whole_memory_grid_equals_three_slabs():
    """1024 x 1024 x 3072 cells on one GPU (242 GB with the fused kernel's two buffer sets; every array
    holds 3.2e9 elements, beyond 32-bit indexing) against the same problem as three 1024-layer slabs, whose
    arrays stay below 2^31 elements: identical checksums of D and B after 5 steps."""
    N, steps = (1024, 1024, 3072), 5
    whole = make(N, (20,) * 6, E.MODE_FUSED, True, steps=steps)
    whole.step(0, steps)
    want = [whole.checksum(i) for i in FIELDS]
    e_whole = whole.sumsq(E.DX)
    whole.close()
    del whole
    assert e_whole > 0 and np.isfinite(e_whole)
    parts = slab.partition(N[2], 3)
    engs = [make(N, (20,) * 6, E.MODE_FUSED, True, z_range=p, steps=steps) for p in parts]
    pl = [[{w: [E.halo_tensor(e, w, c, par) for c in (0, 1)] for w in range(4)} for par in (0, 1)] for e in engs]
    for n in range(steps):
        for e in engs:
            e.fused(n, E.PART_EDGE)
        cur = [pl[r][engs[r].parity()] for r in range(3)]
        for r in range(1, 3):
            for c in (0, 1):
                cur[r - 1][E.HALO_RECV_HI][c].copy_(cur[r][E.HALO_SEND_DOWN][c])
        for e in engs:
            e.fused(n, E.PART_TOP)
        for e in engs:
            e.fused_top()
            e.fused(n, E.PART_INTERIOR)
        for r in range(2):
            for c in (0, 1):
                cur[r + 1][E.HALO_RECV_LO][c].copy_(cur[r][E.HALO_SEND_UP][c])
    got = [sum(e.checksum(i) for e in engs) % 2 ** 64 for i in FIELDS]
    assert got == want


def test_2d_8192_fused_equals_two_kernel_path():
    """8192 x 8192 with 20-layer CPML and the reference's Lorentz slab (the 2-D benchmark workload):
    the one-kernel step and the two-kernel step leave identical Ey, Dy, Bx, Bz (sums in the same
    reduction order) and identical probe series at cells inside, beside and far from the slab."""
    N, steps = 8192, 150
    K = fs.constants()
    dx = K["c0"] / 1.5e14 / 40
    dt = dx / (2 * K["c0"])
    results = []
    for mode in (E.MODE2D_FULL, E.MODE2D_FUSED):
        eng = fdtd_amd.Engine2D(N, N, (20,) * 4, dx, dx, dt, K["mu0"], K["eps0"], N // 2 - 40, N // 2, steps, precision=32, mode=mode)
        fs.load_cpml2d(eng, dx, dx, dt)
        eng.set(E.D2_JY, fs.default_source2d(steps, dx, dx, dt))
        for q, c in enumerate(fs.default_lorentz2d(N, N, dt)):
            eng.set(E.D2_C1 + q, c)
        for cell in ((N // 2, N // 2), (N // 2 - 25, N // 2 + 3), (N // 2 - 60, N // 2 - 40), (10, 4000)):
            eng.probe_add(E.D2_EY, *cell)
        eng.probe_add(E.D2_HZ, N // 2 + 5, N // 2 + 1)
        eng.step(0, steps)
        results.append(([eng.sumsq(i) for i in (E.D2_DY, E.D2_EY, E.D2_BX, E.D2_HX, E.D2_BZ, E.D2_HZ)], eng.probe_read()))
        eng.close()
    assert results[0][0] == results[1][0] and results[0][0][1] > 0
    assert np.array_equal(results[0][1], results[1][1]) and np.abs(results[0][1]).max() > 0
