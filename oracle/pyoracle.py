"""ctypes front end of the test oracle.  TEST INFRASTRUCTURE ONLY.

Two back ends with one interface (``get``/``set``/``step`` on the array ids of
``include/fdtd_engine.h``, numpy float64 arrays in canonical reference order):

* ``Twin3D`` / ``Twin2D``  -- the CPU restatement in ``oracle/yee_twin.hpp`` (fp32 or fp64);
* ``Ref3D`` / ``Ref2D``    -- the unmodified reference solvers compiled into ``oracle/_ref``
  (only where the reference checkout is mounted; ``ref_available()`` tells).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get("FDTD_REFERENCE_ROOT", "/root/reference")

# ---- array ids (mirror of include/fdtd_engine.h) --------------------------------------------
DX, DY, DZ, EX, EY, EZ, BX, BY, BZ, HX, HY, HZ = range(12)
EPS_X, EPS_Y, EPS_Z, MU_X, MU_Y, MU_Z = range(12, 18)
PHI_E, PHI_M, TAB, JX_MAG, JX_WF = 100, 200, 300, 400, 401
FIELDS3D = list(range(12))


def phi_id(magnetic: bool, comp: int, axis: int, side: int) -> int:
    return (PHI_M if magnetic else PHI_E) + (comp * 3 + axis) * 2 + side


def tab_id(axis: int, side: int, kind: int) -> int:
    return TAB + (axis * 2 + side) * 6 + kind


ALL_PHI3D = [phi_id(m, c, a, s) for m in (False, True) for c in range(3) for a in range(3) if a != c for s in range(2)]

D2_DY, D2_EY, D2_EY_N1, D2_EY_N2, D2_SEY, D2_SEY_N1, D2_SEY_N2 = range(7)
D2_BX, D2_HX, D2_HX_N1, D2_HX_N2, D2_SHX, D2_SHX_N1, D2_SHX_N2 = range(7, 14)
D2_BZ, D2_HZ, D2_HZ_N1, D2_HZ_N2, D2_SHZ, D2_SHZ_N1, D2_SHZ_N2 = range(14, 21)
D2_C1, D2_C2, D2_C3 = 21, 22, 23
D2_PHI, D2_TAB, D2_JY = 100, 300, 400
FIELDS2D = list(range(21))
ALL_PHI2D = list(range(100, 108))


def shape3d(id_: int, N, npml) -> tuple:
    """Canonical numpy shape of a 3-D array id."""
    Nx, Ny, Nz = N
    if 0 <= id_ < 18:
        c, fam = id_ % 3, id_ // 3
        electric = fam in (0, 1, 4)
        return tuple(N[a] + ((a != c) if electric else (a == c)) for a in range(3))
    if PHI_E <= id_ < PHI_E + 18 or PHI_M <= id_ < PHI_M + 18:
        mag = id_ >= PHI_M
        r = id_ - (PHI_M if mag else PHI_E)
        side, a, c = r % 2, (r // 2) % 3, r // 6
        s = [N[q] + ((q == c) if mag else (q != c)) for q in range(3)]
        s[a] = npml[a * 2 + side]
        return tuple(s)
    if TAB <= id_ < TAB + 36:
        return (npml[(id_ - TAB) // 6],)
    if id_ == JX_MAG:
        return (Nx, Ny + 1)
    raise KeyError(id_)


def shape2d(id_: int, Nx: int, Nz: int, npml) -> tuple:
    if 0 <= id_ <= 20:
        return [(Nx + 1, Nz + 1), (Nx + 1, Nz), (Nx, Nz + 1)][id_ // 7]
    if id_ in (D2_C1, D2_C2, D2_C3):
        return (Nx, Nz)
    if D2_PHI <= id_ < D2_PHI + 8:
        which, side = (id_ - D2_PHI) // 2, (id_ - D2_PHI) % 2
        if which in (0, 2):
            return (Nx + 1, npml[2 + side])
        return (npml[side], Nz + 1)
    if D2_TAB <= id_ < D2_TAB + 24:
        return (npml[(id_ - D2_TAB) // 6],)
    raise KeyError(id_)


# ---- library loading -------------------------------------------------------------------------
def build_twin() -> str:
    path = os.path.join(HERE, "libyee_twin.so")
    srcs = [os.path.join(HERE, f) for f in ("yee_twin_capi.cpp", "yee_twin.hpp")]
    if not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs):
        subprocess.check_call(["make", "-C", HERE, "twin"], stdout=subprocess.DEVNULL)
    return path


def ref_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "fdtd3d", "src")) or all(
        os.path.exists(os.path.join(HERE, "_ref", f)) for f in ("libref3d.so", "libref2d.so"))


def build_ref() -> None:
    if os.path.isdir(os.path.join(REF_ROOT, "fdtd3d", "src")):
        subprocess.check_call(["make", "-C", HERE, "ref", f"REF={REF_ROOT}"], stdout=subprocess.DEVNULL)


_libs: dict = {}


def _lib(name: str) -> C.CDLL:
    if name not in _libs:
        if name == "twin":
            lib = C.CDLL(build_twin())
        else:
            build_ref()
            lib = C.CDLL(os.path.join(HERE, "_ref", f"lib{name}.so"))
        _libs[name] = lib
    return _libs[name]


_dp = C.POINTER(C.c_double)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(_dp)


class _Base:
    """get/set/step over a C handle; subclasses bind the symbol prefix."""
    _prefix = ""
    _libname = ""

    def _f(self, name, restype=C.c_int):
        fn = getattr(_lib(self._libname), self._prefix + name)
        fn.restype = restype
        return fn

    def size(self, id_: int) -> int:
        return self._f("size", C.c_size_t)(self.h, C.c_int(id_))

    def get(self, id_: int, shape=None) -> np.ndarray:
        n = self.size(id_)
        out = np.zeros(n, dtype=np.float64)
        if n:
            rc = self._f("get")(self.h, C.c_int(id_), _ptr(out), C.c_size_t(n))
            if rc:
                raise RuntimeError(f"get({id_}) failed")
        return out.reshape(shape) if shape is not None else out

    def set(self, id_: int, arr) -> None:
        a = np.ascontiguousarray(arr, dtype=np.float64).ravel()
        if a.size != self.size(id_):
            raise ValueError(f"set({id_}): size {a.size} != {self.size(id_)}")
        if a.size and self._f("set")(self.h, C.c_int(id_), _ptr(a), C.c_size_t(a.size)):
            raise RuntimeError(f"set({id_}) failed")

    def peek(self, id_: int, flat: int) -> float:
        """One entry of a field array by flat canonical index (compiled reference only)."""
        return self._f("peek", C.c_double)(self.h, C.c_int(id_), C.c_size_t(flat))

    def step(self, n0: int, count: int = 1) -> None:
        self._f("step", None)(self.h, C.c_int(n0), C.c_int(count))

    def close(self):
        if getattr(self, "h", None):
            self._f("destroy", None)(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def consts() -> dict:
    out = (C.c_double * 4)()
    _lib("twin").twin_consts(out)
    return dict(c0=out[0], mu0=out[1], eps0=out[2], Z0=out[3])


class Twin3D(_Base):
    _prefix, _libname = "twin3d_", "twin"

    def __init__(self, prec, N, npml, d, dt, mu0, kJ, n_wf, has_mu=False):
        fn = self._f("create", C.c_void_p)
        self.N, self.npml, self.prec = tuple(N), tuple(npml), prec
        self.h = C.c_void_p(fn(C.c_int(prec), *(C.c_int(v) for v in N), (C.c_int * 6)(*npml),
                               *(C.c_double(v) for v in d), C.c_double(dt), C.c_double(mu0),
                               C.c_int(kJ), C.c_int(n_wf), C.c_int(int(has_mu))))
        if not self.h:
            raise ValueError("twin3d_create rejected the arguments")

    def half_e(self, n):
        self._f("half_e", None)(self.h, C.c_int(n))

    def half_h(self):
        self._f("half_h", None)(self.h)

    def observe_accumulate(self, weight, reset):
        self._f("observe_accumulate", None)(self.h, C.c_double(weight), C.c_int(int(reset)))

    def observe_rms(self):
        out = np.zeros(self.N[0] * self.N[1] * self.N[2])
        self._f("observe_rms", None)(self.h, _ptr(out))
        return out

    def observe_frame(self):
        n = self.N[0] * self.N[1] * self.N[2]
        a, b = np.zeros(n), np.zeros(n)
        self._f("observe_frame", None)(self.h, _ptr(a), _ptr(b))
        return a, b

    def fill_materials(self, seed, eps_lo, eps_hi, mu_lo, mu_hi):
        self._f("fill_materials", None)(self.h, C.c_uint64(seed), C.c_double(eps_lo), C.c_double(eps_hi),
                                        C.c_double(mu_lo), C.c_double(mu_hi))


class Twin2D(_Base):
    _prefix, _libname = "twin2d_", "twin"

    def __init__(self, prec, Nx, Nz, npml, dx, dz, dt, mu0, eps0, iJ, kJ, n_wf):
        fn = self._f("create", C.c_void_p)
        self.Nx, self.Nz, self.npml, self.prec = Nx, Nz, tuple(npml), prec
        self.h = C.c_void_p(fn(C.c_int(prec), C.c_int(Nx), C.c_int(Nz), (C.c_int * 4)(*npml), C.c_double(dx),
                               C.c_double(dz), C.c_double(dt), C.c_double(mu0), C.c_double(eps0),
                               C.c_int(iJ), C.c_int(kJ), C.c_int(n_wf)))
        if not self.h:
            raise ValueError("twin2d_create rejected the arguments")

    def observe_accumulate(self, weight, reset):
        self._f("observe_accumulate", None)(self.h, C.c_double(weight), C.c_int(int(reset)))

    def observe_rms(self):
        out = np.zeros(self.Nx * self.Nz)
        self._f("observe_rms", None)(self.h, _ptr(out))
        return out


def cpml_face(n, side, d_step, d_den, dt) -> np.ndarray:
    """[6][n] tables b_e a_e C_e b_m a_m C_m of one CPML face (fp64 restatement)."""
    out = np.zeros((6, max(n, 0)), dtype=np.float64)
    if n > 0:
        fn = _lib("twin").twin_cpml_face
        fn.restype = None
        fn(C.c_int(n), C.c_int(side), C.c_double(d_step), C.c_double(d_den), C.c_double(dt), _ptr(out))
    return out


def default_source3d(Nx, Ny, n_wf, d, dt):
    wf = np.zeros(n_wf); mag = np.zeros((Nx, Ny + 1))
    fn = _lib("twin").twin_default_source3d
    fn.restype = None
    fn(C.c_int(Nx), C.c_int(Ny), C.c_int(n_wf), *(C.c_double(v) for v in d), C.c_double(dt), _ptr(wf), _ptr(mag))
    return wf, mag


def default_materials3d(N, d):
    Nx, Ny, Nz = N
    e = [np.zeros((Nx, Ny + 1, Nz + 1)), np.zeros((Nx + 1, Ny, Nz + 1)), np.zeros((Nx + 1, Ny + 1, Nz))]
    fn = _lib("twin").twin_default_materials3d
    fn.restype = None
    fn(C.c_int(Nx), C.c_int(Ny), C.c_int(Nz), *(C.c_double(v) for v in d), *(_ptr(a) for a in e))
    return e


def default_source2d(n_wf, dx, dz, dt):
    jy = np.zeros(n_wf)
    fn = _lib("twin").twin_default_source2d
    fn.restype = None
    fn(C.c_int(n_wf), C.c_double(dx), C.c_double(dz), C.c_double(dt), _ptr(jy))
    return jy


def default_lorentz2d(Nx, Nz, dt):
    c = [np.zeros((Nx, Nz)) for _ in range(3)]
    fn = _lib("twin").twin_default_lorentz2d
    fn.restype = None
    fn(C.c_int(Nx), C.c_int(Nz), C.c_double(dt), *(_ptr(a) for a in c))
    return c


def hash_unit(seed, stream, index) -> float:
    fn = _lib("twin").twin_hash_unit
    fn.restype = C.c_double
    return fn(C.c_uint64(seed), C.c_uint32(stream), C.c_uint64(index))


def _scratch() -> str:
    d = os.path.join(tempfile.gettempdir(), f"fdtd_ref_scratch_{os.getpid()}")
    os.makedirs(d, exist_ok=True)
    return d


class Ref3D(_Base):
    """The compiled reference FDTD3D, re-parameterised through its own setup functions."""
    _prefix, _libname = "ref3d_", "ref3d"

    def __init__(self, N=None, npml=None, steps=None):
        cwd = os.getcwd()
        try:
            self.h = C.c_void_p(self._f("create", C.c_void_p)(_scratch().encode()))
        finally:
            os.chdir(cwd)
        if not self.h:
            raise RuntimeError("ref3d_create failed")
        if N is not None:
            if self._f("reshape")(self.h, *(C.c_int(v) for v in N), (C.c_int * 6)(*npml), C.c_int(steps)):
                raise ValueError("ref3d_reshape rejected the arguments")
        s = self.scalars()
        self.N = (s["Nx"], s["Ny"], s["Nz"])

    def scalars(self) -> dict:
        d = (C.c_double * 8)(); i = (C.c_int * 8)()
        self._f("scalars", None)(self.h, d, i)
        return dict(dx=d[0], dy=d[1], dz=d[2], dt=d[3], mu0=d[4], eps0=d[5], c0=d[6], Z0=d[7],
                    Nx=i[0], Ny=i[1], Nz=i[2], steps=i[3], k_Jx=i[4])

    def run(self) -> str:
        """Run the reference's own time loop; returns the directory holding its output files."""
        self._f("run", None)(self.h)
        return os.path.join(_scratch(), "output_3d")

    def set_kjx(self, k):
        self._f("set_kjx", None)(self.h, C.c_int(k))

    def tables_consistent(self) -> bool:
        return self._f("tables_consistent")(self.h) == 0

    def half_e(self, n):
        self._f("half_e", None)(self.h, C.c_int(n))

    def half_h(self):
        self._f("half_h", None)(self.h)


class Ref2D(_Base):
    _prefix, _libname = "ref2d_", "ref2d"

    def __init__(self, Nx=None, Nz=None, npml=None, steps=None):
        cwd = os.getcwd()
        try:
            self.h = C.c_void_p(self._f("create", C.c_void_p)(_scratch().encode()))
        finally:
            os.chdir(cwd)
        if not self.h:
            raise RuntimeError("ref2d_create failed")
        if Nx is not None:
            if self._f("reshape")(self.h, C.c_int(Nx), C.c_int(Nz), (C.c_int * 4)(*npml), C.c_int(steps)):
                raise ValueError("ref2d_reshape rejected the arguments")

    def scalars(self) -> dict:
        d = (C.c_double * 8)(); i = (C.c_int * 8)()
        self._f("scalars", None)(self.h, d, i)
        return dict(dx=d[0], dz=d[1], dt=d[2], mu0=d[3], eps0=d[4], c0=d[5],
                    Nx=i[0], Nz=i[1], steps=i[2], i_Jy=i[3], k_Jy=i[4])

    def run(self) -> str:
        self._f("run", None)(self.h)
        return os.path.join(_scratch(), "output_2d")

    def set_source_cell(self, i, k):
        self._f("set_source_cell", None)(self.h, C.c_int(i), C.c_int(k))
