"""Host-side setup generators (include/fdtd_setup.h): constants, CPML tables, default sources and
materials of the reference problems.  Pure host code inside the engine library; usable without a GPU."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .engine import EngineError, lib

_dp = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(_dp)


def constants() -> dict:
    v = [C.c_double() for _ in range(4)]
    lib().fdtd_constants(*(C.byref(x) for x in v))
    return dict(c0=v[0].value, mu0=v[1].value, eps0=v[2].value, Z0=v[3].value)


def cpml_tables(n: int, side: int, d_step: float, d_den: float, dt: float) -> np.ndarray:
    """[6][n]: b_e a_e C_e b_m a_m C_m of one face (side 0 low, 1 high)."""
    out = np.zeros((6, max(n, 0)))
    if n > 0 and lib().fdtd_cpml_tables(C.c_int(n), C.c_int(side), C.c_double(d_step), C.c_double(d_den), C.c_double(dt), _p(out)):
        raise EngineError("fdtd_cpml_tables rejected its arguments")
    return out


def default_source3d(Nx, Ny, n_wf, d, dt):
    wf, mag = np.zeros(n_wf), np.zeros((Nx, Ny + 1))
    if lib().fdtd3d_default_source(C.c_int(Nx), C.c_int(Ny), C.c_int(n_wf), *(C.c_double(v) for v in d), C.c_double(dt), _p(wf), _p(mag)):
        raise EngineError("fdtd3d_default_source rejected its arguments")
    return wf, mag


def default_materials3d(N, d):
    Nx, Ny, Nz = N
    e = [np.zeros((Nx, Ny + 1, Nz + 1)), np.zeros((Nx + 1, Ny, Nz + 1)), np.zeros((Nx + 1, Ny + 1, Nz))]
    if lib().fdtd3d_default_materials(C.c_int(Nx), C.c_int(Ny), C.c_int(Nz), *(C.c_double(v) for v in d), *(_p(a) for a in e)):
        raise EngineError("fdtd3d_default_materials rejected its arguments")
    return e


def default_source2d(n_wf, dx, dz, dt):
    jy = np.zeros(n_wf)
    if lib().fdtd2d_default_source(C.c_int(n_wf), C.c_double(dx), C.c_double(dz), C.c_double(dt), _p(jy)):
        raise EngineError("fdtd2d_default_source rejected its arguments")
    return jy


def default_lorentz2d(Nx, Nz, dt):
    c = [np.zeros((Nx, Nz)) for _ in range(3)]
    if lib().fdtd2d_default_lorentz(C.c_int(Nx), C.c_int(Nz), C.c_double(dt), *(_p(a) for a in c)):
        raise EngineError("fdtd2d_default_lorentz rejected its arguments")
    return c


def load_cpml3d(engine, d, dt) -> None:
    """Generate and upload the tables of all six faces of a 3-D engine."""
    from .engine import TAB
    for a in range(3):
        for s in range(2):
            n = engine.npml[a * 2 + s]
            if n:
                t = cpml_tables(n, s, d[a], d[a], dt)
                for k in range(6):
                    engine.set(TAB + (a * 2 + s) * 6 + k, t[k])


def load_cpml2d(engine, dx, dz, dt) -> None:
    from .engine import D2_TAB
    dd = (dx, dz)
    for a in range(2):
        for s in range(2):
            n = engine.npml[a * 2 + s]
            if n:
                t = cpml_tables(n, s, dd[a], dz, dt)       # the x faces divide by dz, as the reference does
                for k in range(6):
                    engine.set(D2_TAB + (a * 2 + s) * 6 + k, t[k])
