#!/usr/bin/env python3
"""Round-off drift of the fp32 engine against the fp64 engine over the reference's own run lengths:
    python tools/fp32_drift.py [3d|2d|cavity]
The fp64 engine equals the compiled reference bit for bit (tests/test_gpu_full_run.py), so the numbers
are the fp32 engine's relative L2 distance from the reference, per vector field, as a function of the
step.  3d: the reference's default 200^3 problem, 1501 steps (FDTD3D.cpp:39); 2d: its default 200x480
Lorentz-slab problem, 1580 steps (FDTD2D.cpp:38); cavity: 128^3 lossless vacuum cavity with a dipole."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd_amd  # noqa: E402
from fdtd_amd import engine as E  # noqa: E402
from fdtd_amd import setup as fs  # noqa: E402


def fam(a, b, ids):
    num = sum(float(np.sum((a.get(i) - b.get(i)) ** 2)) for i in ids)
    den = sum(float(np.sum(b.get(i) ** 2)) for i in ids)
    return (num / den) ** 0.5 if den > 0 else 0.0


def run3d(cavity=False):
    K = fs.constants()
    d = (1053e-9 / 20,) * 3
    dt = d[0] / (2 * K["c0"])
    N, steps, npml, kJ = ((128,) * 3, 1501, (0,) * 6, 64) if cavity else ((200,) * 3, 1501, (20,) * 6, 20)
    engs = []
    for prec in (64, 32):
        e = fdtd_amd.Engine3D(N, npml, d, dt, K["mu0"], kJ, steps, precision=prec, mode=E.MODE_FUSED)
        if not cavity:
            fs.load_cpml3d(e, d, dt)
        wf, mag = fs.default_source3d(N[0], N[1], steps, d, dt)
        if cavity:
            mag = np.zeros_like(mag)
            mag[N[0] // 2, N[1] // 2] = 1.0
            for c in range(3):
                e.set(E.EPS_X + c, np.full(e.size(E.EPS_X + c), K["eps0"]))
        else:
            e.fill_reference_materials()
        e.set(E.JX_WF, wf)
        e.set(E.JX_MAG, mag)
        engs.append(e)
    ref, low = engs
    print(f"| step | rel-L2 E | rel-L2 H | rel-L2 D | rel-L2 B | 6e-8*sqrt(step) |\n|---|---|---|---|---|---|")
    n = 0
    for stop in [10, 40, 120, 200] + list(range(300, 1501, 100)) + [1501]:
        for e in engs:
            e.step(n, stop - n)
        n = stop
        print(f"| {n} | {fam(low, ref, (E.EX, E.EY, E.EZ)):.2e} | {fam(low, ref, (E.HX, E.HY, E.HZ)):.2e} | "
              f"{fam(low, ref, (E.DX, E.DY, E.DZ)):.2e} | {fam(low, ref, (E.BX, E.BY, E.BZ)):.2e} | {6e-8 * n ** 0.5:.2e} |", flush=True)


def run2d():
    K = fs.constants()
    Nx, Nz, steps = 200, 480, 1580
    lam = K["c0"] / 1.5e14
    dx = dz = lam / 40
    dt = dx / (2 * K["c0"])
    engs = []
    for prec in (64, 32):
        e = fdtd_amd.Engine2D(Nx, Nz, (20,) * 4, dx, dz, dt, K["mu0"], K["eps0"], Nx // 2 - 40, Nz // 2, steps, precision=prec, mode=E.MODE2D_FUSED)
        fs.load_cpml2d(e, dx, dz, dt)
        for id_, c in zip((E.D2_C1, E.D2_C2, E.D2_C3), fs.default_lorentz2d(Nx, Nz, dt)):
            e.set(id_, c)
        e.set(E.D2_JY, fs.default_source2d(steps, dx, dz, dt))
        engs.append(e)
    ref, low = engs
    print("| step | rel-L2 Ey | rel-L2 (Hx,Hz) | 6e-8*sqrt(step) |\n|---|---|---|---|")
    n = 0
    for stop in [10, 40, 160] + list(range(200, 1580, 100)) + [1580]:
        for e in engs:
            e.step(n, stop - n)
        n = stop
        print(f"| {n} | {fam(low, ref, (E.D2_EY,)):.2e} | {fam(low, ref, (E.D2_HX, E.D2_HZ)):.2e} | {6e-8 * n ** 0.5:.2e} |", flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "3d"
    {"3d": run3d, "2d": run2d, "cavity": lambda: run3d(True)}[what]()
