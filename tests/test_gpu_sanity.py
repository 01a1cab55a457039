"""The reference's own after-the-fact checks (fdtd3d/scripts/verify_vti.py:140-148, :255-262 and
diagnose_energy.py:19-33) as automated guards on its default 200^3 problem, run through the fused fp32
path: every |E| frame finite, energy growth between frames within [0.1, 100], energy reaching the
centre of the grid, and the 20-cell rim (the CPML) not holding the bulk of it."""
import numpy as np
import pytest

import fdtd_amd
from fdtd_amd import engine as E
from fdtd_amd import setup as fs

pytestmark = pytest.mark.gpu


def test_reference_default_problem_passes_its_own_heuristics():
    N, steps = (200, 200, 200), 520
    K = fs.constants()
    d = (1053e-9 / 20,) * 3
    dt = d[0] / (2 * K["c0"])
    eng = fdtd_amd.Engine3D(N, (20,) * 6, d, dt, K["mu0"], 20, steps, precision=32, mode=E.MODE_FUSED)
    fs.load_cpml3d(eng, d, dt)
    wf, mag = fs.default_source3d(N[0], N[1], steps, d, dt)
    eng.set(E.JX_WF, wf)
    eng.set(E.JX_MAG, mag)
    eng.fill_reference_materials()
    energies = []
    n = 0
    for frame in (400, 460, 520):
        eng.step(n, frame - n)
        n = frame
        _, mag_e = eng.observe_frame()
        vol = mag_e.reshape(N[2], N[1], N[0])                     # [z][y][x], as the VTI files hold it
        assert np.all(np.isfinite(vol))
        dens = vol ** 2
        total = dens.sum()
        centre = dens[90:110, 90:110, 90:110].sum()
        rim = (dens[:20].sum() + dens[-20:].sum() + dens[:, :20].sum() + dens[:, -20:].sum() + dens[:, :, :20].sum() + dens[:, :, -20:].sum())
        assert total > 0 and centre > 0
        assert rim / total < 0.5
        energies.append(total)
    for a, b in zip(energies, energies[1:]):
        assert 0.1 < b / a < 100
