"""ctypes binding of the engine's C ABI (include/fdtd_engine.h).

Mirrors the reference's solver interface for the hot path: an ``Engine3D`` stands where
``FDTD3D``'s private state plus ``updateE``/``updateH`` stood (fdtd3d/include/FDTD3D.h:17-195),
an ``Engine2D`` where ``FDTD2D``'s state plus ``stepOnce`` stood (fdtd2d/include/FDTD2D.h:16-89).
Arrays cross as float64 numpy arrays in the reference's own flat order.  There is no CPU
fallback: without the built library (``fdtd_amd/libfdtd_engine.so``) or without a HIP device
every constructor raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FDTD_ENGINE_LIB: another build of the same library (kernel experiments); the product path is the in-tree file
LIB_PATH = os.environ.get("FDTD_ENGINE_LIB") or os.path.join(_HERE, "libfdtd_engine.so")

# array ids (include/fdtd_engine.h)
DX, DY, DZ, EX, EY, EZ, BX, BY, BZ, HX, HY, HZ = range(12)
EPS_X, EPS_Y, EPS_Z, MU_X, MU_Y, MU_Z = range(12, 18)
PHI_E, PHI_M, TAB, JX_MAG, JX_WF = 100, 200, 300, 400, 401
MODE_FULL, MODE_LEAN, MODE_FUSED = 0, 1, 2
MODE2D_FULL, MODE2D_FUSED = 0, 1
PART_ALL, PART_EDGE, PART_INTERIOR, PART_TOP = 0, 1, 2, 3
HALO_SEND_UP, HALO_RECV_LO, HALO_SEND_DOWN, HALO_RECV_HI = 0, 1, 2, 3

D2_DY, D2_EY, D2_EY_N1, D2_EY_N2, D2_SEY, D2_SEY_N1, D2_SEY_N2 = range(7)
D2_BX, D2_HX = 7, 8
D2_BZ, D2_HZ = 14, 15
D2_C1, D2_C2, D2_C3 = 21, 22, 23
D2_PHI, D2_TAB, D2_JY = 100, 300, 400


def kernel_source_digest() -> str:
    """sha256 over the kernel sources the library was built from, so that a measurement taken on one build
    (HBM traffic from a profiler pass) is not quoted next to timings of another."""
    import hashlib
    h = hashlib.sha256()
    for name in ("common.hpp", "kernels3d.hpp", "kernels3d_fused.hpp", "kernels3d_pair.hpp", "kernels2d.hpp", "kernels2d_fused.hpp", "kernels2d_pair.hpp"):
        with open(os.path.join(_HERE, "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


class EngineError(RuntimeError):
    pass


class Desc3D(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("Nx", C.c_int32), ("Ny", C.c_int32), ("Nz", C.c_int32),
                ("n_pml", C.c_int32 * 6), ("k_Jx", C.c_int32), ("n_wf", C.c_int32), ("precision", C.c_int32),
                ("has_mu", C.c_int32), ("mode", C.c_int32), ("z_lo", C.c_int32), ("z_hi", C.c_int32),
                ("device", C.c_int32), ("dx", C.c_double), ("dy", C.c_double), ("dz", C.c_double),
                ("dt", C.c_double), ("mu0", C.c_double)]


class Desc2D(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("Nx", C.c_int32), ("Nz", C.c_int32), ("n_pml", C.c_int32 * 4),
                ("i_Jy", C.c_int32), ("k_Jy", C.c_int32), ("n_wf", C.c_int32), ("precision", C.c_int32),
                ("device", C.c_int32), ("dx", C.c_double), ("dz", C.c_double), ("dt", C.c_double),
                ("mu0", C.c_double), ("eps0", C.c_double), ("mode", C.c_int32), ("reserved", C.c_int32)]


_lib = None
_dp = C.POINTER(C.c_double)


def lib() -> C.CDLL:
    """Load the shared library; raise if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EngineError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); there is no fallback path")
        # PyTorch ships its own copy of the HIP runtime; a process that hands device memory to
        # torch.distributed must not end up with two of them, so torch (if present) loads its copy first
        # and this library then binds to that one.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.fdtd_last_error.restype = C.c_char_p
        L.fdtd_engine_version.restype = C.c_char_p
        for name in ("fdtd3d_array_size", "fdtd3d_device_bytes", "fdtd2d_array_size", "fdtd2d_device_bytes"):
            getattr(L, name).restype = C.c_size_t
        _lib = L
    return _lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise EngineError(f"{what}: {lib().fdtd_last_error().decode()} (status {rc})")


def _stream_handle(stream) -> C.c_void_p:
    if stream is None:
        return C.c_void_p(0)
    if hasattr(stream, "cuda_stream"):       # torch.cuda.Stream
        return C.c_void_p(stream.cuda_stream)
    return C.c_void_p(int(stream))


class _Common:
    _p = ""

    def _fn(self, name):
        return getattr(lib(), self._p + name)

    def size(self, id_: int) -> int:
        return self._fn("array_size")(self.h, C.c_int(id_))

    def set(self, id_: int, arr) -> None:
        a = np.ascontiguousarray(arr, dtype=np.float64).ravel()
        _check(self._fn("set_array")(self.h, C.c_int(id_), a.ctypes.data_as(_dp), C.c_size_t(a.size)), f"set_array({id_})")

    def get(self, id_: int, shape=None, out=None) -> np.ndarray:
        n = self.size(id_)
        if out is None:
            out = np.zeros(n, dtype=np.float64)
        flat = out.reshape(-1)
        _check(self._fn("get_array")(self.h, C.c_int(id_), flat.ctypes.data_as(_dp), C.c_size_t(flat.size)), f"get_array({id_})")
        return out.reshape(shape) if shape is not None else out

    def sumsq(self, id_: int, stream=None) -> float:
        v = C.c_double(0)
        _check(self._fn("sumsq")(self.h, C.c_int(id_), C.byref(v), _stream_handle(stream)), f"sumsq({id_})")
        return v.value

    def device_bytes(self) -> int:
        return self._fn("device_bytes")(self.h)

    # ---- probe time series (samples stay on the device until read) ----
    def probe_add(self, id_: int, *cell) -> None:
        """Sample array `id_` at canonical cell (i, j, k) [3-D] or (i, k) [2-D] after every step."""
        _check(self._fn("probe_add")(self.h, C.c_int(id_), *[C.c_int(int(c)) for c in cell]), f"probe_add({id_}, {cell})")
        self.probe_count = getattr(self, "probe_count", 0) + 1

    def probe_record(self, stream=None) -> None:
        _check(self._fn("probe_record")(self.h, _stream_handle(stream)), "probe_record")

    def probe_read(self) -> np.ndarray:
        """[samples, probes] array of everything recorded so far."""
        ns, npr = C.c_int(0), C.c_int(0)
        _check(self._fn("probe_read")(self.h, None, C.c_size_t(0), C.byref(ns), C.byref(npr)), "probe_read")
        out = np.zeros((ns.value, npr.value))
        if out.size:
            _check(self._fn("probe_read")(self.h, out.ctypes.data_as(_dp), C.c_size_t(out.size), C.byref(ns), C.byref(npr)), "probe_read")
        return out

    def probe_reset(self, drop_probes: bool = False) -> None:
        _check(self._fn("probe_reset")(self.h, C.c_int(int(drop_probes))), "probe_reset")
        if drop_probes:
            self.probe_count = 0

    def kernel_timing(self, enable: bool) -> None:
        _check(self._fn("kernel_timing")(self.h, C.c_int(int(enable))), "kernel_timing")

    def kernel_time(self, kind: int):
        """(launches, total device milliseconds) of kernel kind 0 (E) or 1 (H) since the last call."""
        n, ms = C.c_int(0), C.c_double(0)
        _check(self._fn("kernel_time")(self.h, C.c_int(kind), C.byref(n), C.byref(ms)), "kernel_time")
        return n.value, ms.value

    def close(self) -> None:
        if getattr(self, "h", None):
            self._fn("destroy")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine3D(_Common):
    """Device-resident 3-D Yee state of one z-slab plus its two half-step kernels."""
    _p = "fdtd3d_"

    def __init__(self, N, npml, d, dt, mu0, k_Jx, n_wf, precision=32, has_mu=False, mode=MODE_FULL,
                 z_range=None, device=-1):
        self.N, self.npml, self.precision = tuple(N), tuple(npml), precision
        z_lo, z_hi = z_range if z_range is not None else (0, N[2])
        desc = Desc3D(C.sizeof(Desc3D), N[0], N[1], N[2], (C.c_int32 * 6)(*npml), k_Jx, n_wf, precision, int(has_mu),
                      mode, z_lo, z_hi, device, d[0], d[1], d[2], dt, mu0)
        self.desc = desc
        self.h = C.c_void_p()
        _check(lib().fdtd3d_create(C.byref(desc), C.byref(self.h)), "fdtd3d_create")

    def step(self, n0: int, count: int = 1, stream=None) -> None:
        _check(lib().fdtd3d_step(self.h, C.c_int(n0), C.c_int(count), _stream_handle(stream)), "fdtd3d_step")

    def step_pairs(self, n0: int, pairs: int = 1, stream=None) -> None:
        """Steps n0 .. n0+2*pairs-1, two per launch (fdtd3d_step_pairs: FUSED mode; probes only on a whole-grid engine)."""
        _check(lib().fdtd3d_step_pairs(self.h, C.c_int(n0), C.c_int(pairs), _stream_handle(stream)), "fdtd3d_step_pairs")

    def planes(self, id_: int, k: int, count: int = 1):
        """(device pointer, bytes) of `count` stored planes of array `id_` from global plane k, current buffers."""
        p, n = C.c_void_p(), C.c_size_t()
        _check(lib().fdtd3d_planes(self.h, C.c_int(id_), C.c_int(k), C.c_int(count), C.byref(p), C.byref(n)), "fdtd3d_planes")
        return p.value, n.value

    def reduce_range(self, id_: int, k_lo: int, k_hi: int, stream=None):
        """(sum of squares, checksum) of array `id_` over the owned planes inside the global range [k_lo, k_hi)."""
        ss, cs = C.c_double(), C.c_uint64()
        _check(lib().fdtd3d_reduce_range(self.h, C.c_int(id_), C.c_int(k_lo), C.c_int(k_hi), C.byref(ss), C.byref(cs),
                                         _stream_handle(stream)), "fdtd3d_reduce_range")
        return ss.value, cs.value

    def pair_info(self) -> dict:
        """Whether step() goes two steps per launch on this engine, and the two-step kernel's launch shape."""
        o = (C.c_int * 4)()
        _check(lib().fdtd3d_pair_info(self.h, o), "fdtd3d_pair_info")
        info = dict(zip(("used_by_step", "planes_per_group", "rows_per_group", "elements_per_lane"), list(o)))
        _check(lib().fdtd3d_pair_layout(self.h, o), "fdtd3d_pair_layout")
        info.update(zip(("x_tiles", "tile0_left_lanes", "tile0_right_from", "xcd_run"), list(o)))
        return info

    def stamp_shape(self) -> dict:
        """Everything about the launch shapes a profiler pass has to share with a timed run to be quoted beside it."""
        shape = self.launch_shape()
        pair = self.pair_info()
        if pair["used_by_step"]:
            shape["pair"] = pair
        return shape

    def pair_part(self, n: int, part: int = PART_ALL, stream=None) -> None:
        """Steps n, n+1 in one launch (PART_ALL) or as PART_EDGE then PART_INTERIOR (fdtd3d_pair_part)."""
        _check(lib().fdtd3d_pair_part(self.h, C.c_int(n), C.c_int(part), _stream_handle(stream)), "fdtd3d_pair_part")

    def pair_shape(self, chunk: int = 0, rows: int = 0) -> None:
        _check(lib().fdtd3d_pair_shape(self.h, C.c_int(chunk), C.c_int(rows)), "fdtd3d_pair_shape")

    def half_e(self, n: int, part: int = PART_ALL, stream=None) -> None:
        _check(lib().fdtd3d_half_e(self.h, C.c_int(n), C.c_int(part), _stream_handle(stream)), "fdtd3d_half_e")

    def half_h(self, part: int = PART_ALL, stream=None) -> None:
        _check(lib().fdtd3d_half_h(self.h, C.c_int(part), _stream_handle(stream)), "fdtd3d_half_h")

    def fused(self, n: int, part: int = PART_ALL, stream=None) -> None:
        """Both halves of step n in one kernel (MODE_FUSED engines); see fdtd3d_fused in the header."""
        _check(lib().fdtd3d_fused(self.h, C.c_int(n), C.c_int(part), _stream_handle(stream)), "fdtd3d_fused")

    def fused_top(self, stream=None) -> None:
        """Finish B of the top owned layer after the upper neighbour's D has arrived (no-op on the last slab)."""
        _check(lib().fdtd3d_fused_top(self.h, _stream_handle(stream)), "fdtd3d_fused_top")

    def spread_gaps(self):
        """GiB of physical memory left out in front of every large array but the first (empty: one plain allocation)."""
        buf = (C.c_double * 64)()
        n = lib().fdtd3d_spread_gaps(self.h, buf, C.c_int(64))
        return [buf[i] for i in range(min(n, 64))]

    def kernel_timeline(self):
        """[(kind, part, start_ms, duration_ms)] of the launches timed since the last kernel_time() of their kind."""
        n = C.c_int(0)
        _check(lib().fdtd3d_kernel_timeline(self.h, C.c_int(0), None, None, None, None, C.byref(n)), "kernel_timeline")
        cap = n.value
        if cap == 0:
            return []
        kinds, parts = (C.c_int * cap)(), (C.c_int * cap)()
        t0, dur = (C.c_double * cap)(), (C.c_double * cap)()
        _check(lib().fdtd3d_kernel_timeline(self.h, C.c_int(cap), kinds, parts, t0, dur, C.byref(n)), "kernel_timeline")
        return [(kinds[q], parts[q], t0[q], dur[q]) for q in range(min(cap, n.value))]

    def spread_gap_gib(self) -> float:
        """GiB of physical memory left out between two field arrays (0.0: the engine's memory is one plain allocation)."""
        f = lib().fdtd3d_spread_gap_gib
        f.restype = C.c_double
        return float(f(self.h))

    def parity(self) -> int:
        """Which of the two D/B buffer sets is current (changes with every fused step)."""
        return lib().fdtd3d_buffer_parity(self.h)

    @property
    def is_fused(self) -> bool:
        return self.desc.mode == MODE_FUSED

    def fill_materials(self, seed, eps_lo, eps_hi, mu_lo, mu_hi) -> None:
        _check(lib().fdtd3d_fill_materials(self.h, C.c_uint64(seed), C.c_double(eps_lo), C.c_double(eps_hi),
                                           C.c_double(mu_lo), C.c_double(mu_hi)), "fdtd3d_fill_materials")

    def fill_array(self, id_: int, seed: int, lo: float, hi: float) -> None:
        """Counter-hash values in [lo, hi) into a resident field array (a state without zeros)."""
        _check(lib().fdtd3d_fill_array(self.h, C.c_int(id_), C.c_uint64(seed), C.c_double(lo), C.c_double(hi)), "fdtd3d_fill_array")

    def fill_reference_materials(self) -> None:
        """The reference's default eps (glass with an air sphere) generated on the device."""
        _check(lib().fdtd3d_fill_reference_materials(self.h), "fdtd3d_fill_reference_materials")

    def halo(self, which: int, comp: int, parity=None):
        """(device pointer, bytes) of one ghost-exchange plane; `parity` selects the buffer set of a
        MODE_FUSED engine (default: the current one)."""
        p, n = C.c_void_p(), C.c_size_t()
        if parity is None:
            _check(lib().fdtd3d_halo(self.h, C.c_int(which), C.c_int(comp), C.byref(p), C.byref(n)), "fdtd3d_halo")
        else:
            _check(lib().fdtd3d_halo_at(self.h, C.c_int(parity), C.c_int(which), C.c_int(comp), C.byref(p), C.byref(n)), "fdtd3d_halo_at")
        return p.value, n.value

    def observe_accumulate(self, weight: float, reset: bool, stream=None) -> None:
        _check(lib().fdtd3d_observe_accumulate(self.h, C.c_double(weight), C.c_int(int(reset)), _stream_handle(stream)), "fdtd3d_observe_accumulate")

    def observe_rms(self, out=None) -> np.ndarray:
        n = self.N[0] * self.N[1] * self.N[2]
        out = np.zeros(n) if out is None else out
        _check(lib().fdtd3d_observe_rms(self.h, out.ctypes.data_as(_dp), C.c_size_t(n)), "fdtd3d_observe_rms")
        return out

    def step_observed(self, n0: int, count: int, weight: float, window: int, stream=None) -> None:
        """`count` steps, the RMS window advanced after each (fdtd3d_step_observed)."""
        _check(lib().fdtd3d_step_observed(self.h, C.c_int(n0), C.c_int(count), C.c_double(weight), C.c_int(window), _stream_handle(stream)),
               "fdtd3d_step_observed")

    def observe_keep_window(self) -> None:
        _check(lib().fdtd3d_observe_keep_window(self.h), "fdtd3d_observe_keep_window")

    def observe_rms_window(self, out=None) -> np.ndarray:
        """sqrt of the last completed window."""
        n = self.N[0] * self.N[1] * self.N[2]
        out = np.zeros(n) if out is None else out
        _check(lib().fdtd3d_observe_rms_window(self.h, out.ctypes.data_as(_dp), C.c_size_t(n)), "fdtd3d_observe_rms_window")
        return out

    def observe_frame(self, stream=None):
        n = self.N[0] * self.N[1] * self.N[2]
        a, b = np.zeros(n), np.zeros(n)
        _check(lib().fdtd3d_observe_frame(self.h, a.ctypes.data_as(_dp), b.ctypes.data_as(_dp), C.c_size_t(n), _stream_handle(stream)), "fdtd3d_observe_frame")
        return a, b

    def tune(self, chunk: int = 0, remap: int = -1, rows: int = 0) -> None:
        _check(lib().fdtd3d_tune(self.h, C.c_int(chunk), C.c_int(remap), C.c_int(rows)), "fdtd3d_tune")

    def launch_shape(self) -> dict:
        """Launch shape of the step kernels in effect (fdtd3d_launch_shape)."""
        o = (C.c_int * 6)()
        _check(lib().fdtd3d_launch_shape(self.h, o), "fdtd3d_launch_shape")
        return dict(zip(("planes_per_group", "rows_per_group", "xcd_run", "elements_per_lane", "edge_planes", "row_pitch"), list(o)))

    def checksum(self, id_: int, stream=None) -> int:
        v = C.c_uint64(0)
        _check(lib().fdtd3d_checksum(self.h, C.c_int(id_), C.byref(v), _stream_handle(stream)), f"checksum({id_})")
        return v.value


class _Member(Engine3D):
    """A slab of an EngineGroup3D: the group owns the handle."""

    def __init__(self, handle, N, npml, precision, desc):
        self.h, self.N, self.npml, self.precision, self.desc = handle, tuple(N), tuple(npml), precision, desc

    def close(self) -> None:
        self.h = None


class EngineGroup3D:
    """Several z-slabs of one problem, possibly on several GPUs, stepped from this one host thread
    (fdtd3d_group_* in include/fdtd_engine.h).  Arrays are the same global host arrays an Engine3D takes."""

    def __init__(self, N, npml, d, dt, mu0, k_Jx, n_wf, n_slabs, precision=32, has_mu=False, mode=MODE_FUSED, devices=None):
        self.N, self.npml, self.precision = tuple(N), tuple(npml), precision
        desc = Desc3D(C.sizeof(Desc3D), N[0], N[1], N[2], (C.c_int32 * 6)(*npml), k_Jx, n_wf, precision, int(has_mu),
                      mode, 0, N[2], -1, d[0], d[1], d[2], dt, mu0)
        self.desc = desc
        self.h = C.c_void_p()
        L = lib()
        L.fdtd3d_group_member.restype = C.c_void_p
        dev = (C.c_int * n_slabs)(*devices) if devices is not None else None
        _check(L.fdtd3d_group_create(C.byref(desc), C.c_int(n_slabs), dev, C.byref(self.h)), "fdtd3d_group_create")
        self.members = [_Member(C.c_void_p(L.fdtd3d_group_member(self.h, C.c_int(r))), N, npml, precision, desc)
                        for r in range(L.fdtd3d_group_slabs(self.h))]

    def size(self, id_: int) -> int:
        return self.members[0].size(id_)

    def set(self, id_: int, arr) -> None:
        a = np.ascontiguousarray(arr, dtype=np.float64).ravel()
        if id_ >= TAB:                      # tables, source plane and waveform are not split: every slab holds them
            for m in self.members:
                m.set(id_, a)
            return
        _check(lib().fdtd3d_group_set_array(self.h, C.c_int(id_), a.ctypes.data_as(_dp), C.c_size_t(a.size)), f"group_set_array({id_})")

    def get(self, id_: int, shape=None) -> np.ndarray:
        out = np.zeros(self.size(id_), dtype=np.float64)
        _check(lib().fdtd3d_group_get_array(self.h, C.c_int(id_), out.ctypes.data_as(_dp), C.c_size_t(out.size)), f"group_get_array({id_})")
        return out.reshape(shape) if shape is not None else out

    def step(self, n0: int, count: int = 1) -> None:
        _check(lib().fdtd3d_group_step(self.h, C.c_int(n0), C.c_int(count)), "fdtd3d_group_step")

    def sync(self) -> None:
        _check(lib().fdtd3d_group_sync(self.h), "fdtd3d_group_sync")

    def fill_materials(self, seed, eps_lo, eps_hi, mu_lo, mu_hi) -> None:
        _check(lib().fdtd3d_group_fill_materials(self.h, C.c_uint64(seed), C.c_double(eps_lo), C.c_double(eps_hi),
                                                 C.c_double(mu_lo), C.c_double(mu_hi)), "fdtd3d_group_fill_materials")

    def fill_reference_materials(self) -> None:
        _check(lib().fdtd3d_group_fill_reference_materials(self.h), "fdtd3d_group_fill_reference_materials")

    def observe_accumulate(self, weight: float, reset: bool) -> None:
        _check(lib().fdtd3d_group_observe_accumulate(self.h, C.c_double(weight), C.c_int(int(reset))), "fdtd3d_group_observe_accumulate")

    def observe_rms(self) -> np.ndarray:
        out = np.zeros(self.N[0] * self.N[1] * self.N[2])
        _check(lib().fdtd3d_group_observe_rms(self.h, out.ctypes.data_as(_dp), C.c_size_t(out.size)), "fdtd3d_group_observe_rms")
        return out

    def step_observed(self, n0: int, count: int, weight: float, window: int) -> None:
        _check(lib().fdtd3d_group_step_observed(self.h, C.c_int(n0), C.c_int(count), C.c_double(weight), C.c_int(window)), "fdtd3d_group_step_observed")

    def observe_rms_window(self) -> np.ndarray:
        out = np.zeros(self.N[0] * self.N[1] * self.N[2])
        _check(lib().fdtd3d_group_observe_rms_window(self.h, out.ctypes.data_as(_dp), C.c_size_t(out.size)), "fdtd3d_group_observe_rms_window")
        return out

    def observe_frame(self):
        n = self.N[0] * self.N[1] * self.N[2]
        a, b = np.zeros(n), np.zeros(n)
        _check(lib().fdtd3d_group_observe_frame(self.h, a.ctypes.data_as(_dp), b.ctypes.data_as(_dp), C.c_size_t(n)), "fdtd3d_group_observe_frame")
        return a, b

    def _own_reductions(self, id_: int):
        """Per slab, (sum of squares, checksum) over the layers it answers for (slabs that take two steps per launch
        store one layer more either side of a cut)."""
        self.sync()
        Nz = self.N[2]
        return [m.reduce_range(id_, lo, Nz + 1 if hi == Nz else hi) for m, (lo, hi) in zip(self.members, self.bounds())]

    def sumsq(self, id_: int) -> float:
        return sum(ss for ss, _ in self._own_reductions(id_))

    def checksum(self, id_: int) -> int:
        return sum(cs for _, cs in self._own_reductions(id_)) % 2 ** 64

    @property
    def pairs(self) -> bool:
        """Whether step() takes two steps per launch (overlapping slabs)."""
        return bool(lib().fdtd3d_group_pairs(self.h))

    def owner(self, k: int) -> "_Member":
        """The member that owns cell layer k (the last one also plane Nz)."""
        bounds = self.bounds()
        for m, (lo, hi) in zip(self.members, bounds):
            if lo <= k < hi or (hi == self.N[2] and k == hi):
                return m
        raise EngineError(f"no slab owns layer {k}")

    def bounds(self):
        R, Nz = len(self.members), self.N[2]
        base, extra = divmod(Nz, R)
        out, lo = [], 0
        for r in range(R):
            hi = lo + base + (1 if r < extra else 0)
            out.append((lo, hi))
            lo = hi
        return out

    def close(self) -> None:
        if getattr(self, "h", None):
            lib().fdtd3d_group_destroy(self.h)
            self.h = None
            for m in self.members:
                m.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine2D(_Common):
    """Device-resident 2-D (Ey, Hx, Hz) state with Lorentz histories plus its two kernels."""
    _p = "fdtd2d_"

    def __init__(self, Nx, Nz, npml, dx, dz, dt, mu0, eps0, i_Jy, k_Jy, n_wf, precision=32, device=-1, mode=0):
        self.Nx, self.Nz, self.npml, self.precision = Nx, Nz, tuple(npml), precision
        desc = Desc2D(C.sizeof(Desc2D), Nx, Nz, (C.c_int32 * 4)(*npml), i_Jy, k_Jy, n_wf, precision, device,
                      dx, dz, dt, mu0, eps0, mode, 0)
        self.desc = desc
        self.h = C.c_void_p()
        _check(lib().fdtd2d_create(C.byref(desc), C.byref(self.h)), "fdtd2d_create")

    def observe_accumulate(self, weight: float, reset: bool, stream=None) -> None:
        _check(lib().fdtd2d_observe_accumulate(self.h, C.c_double(weight), C.c_int(int(reset)), _stream_handle(stream)), "fdtd2d_observe_accumulate")

    def observe_rms(self) -> np.ndarray:
        out = np.zeros(self.Nx * self.Nz)
        _check(lib().fdtd2d_observe_rms(self.h, out.ctypes.data_as(_dp), C.c_size_t(out.size)), "fdtd2d_observe_rms")
        return out

    def step(self, n0: int, count: int = 1, stream=None) -> None:
        _check(lib().fdtd2d_step(self.h, C.c_int(n0), C.c_int(count), _stream_handle(stream)), "fdtd2d_step")

    def step_observed(self, n0: int, count: int, weight: float, window: int, stream=None) -> None:
        """`count` steps with the RMS window of Ey advanced after each (fdtd2d_step_observed)."""
        _check(lib().fdtd2d_step_observed(self.h, C.c_int(n0), C.c_int(count), C.c_double(weight), C.c_int(window), _stream_handle(stream)),
               "fdtd2d_step_observed")

    def observe_rms_window(self) -> np.ndarray:
        """sqrt of the last completed window, [Nx][Nz] with k fastest."""
        out = np.zeros(self.Nx * self.Nz)
        _check(lib().fdtd2d_observe_rms_window(self.h, out.ctypes.data_as(_dp), C.c_size_t(out.size)), "fdtd2d_observe_rms_window")
        return out


class _DeviceSpan:
    """Exposes a raw device address range through __cuda_array_interface__ so that torch can wrap it
    without copying (torch is only plumbing here: device tensors for torch.distributed)."""

    def __init__(self, ptr: int, nbytes: int, itemsize: int):
        self.__cuda_array_interface__ = {"shape": (nbytes // itemsize,), "typestr": "<f4" if itemsize == 4 else "<f8",
                                         "data": (ptr, False), "version": 2}


def planes_tensor(engine: "Engine3D", id_: int, k: int, count: int = 1):
    """torch tensor aliasing `count` stored planes of array `id_` from global plane k (current buffers; no copy)."""
    import torch
    ptr, nbytes = engine.planes(id_, k, count)
    itemsize = 4 if engine.precision == 32 else 8
    return torch.as_tensor(_DeviceSpan(ptr, nbytes, itemsize), device=torch.device("cuda", torch.cuda.current_device()))


def halo_tensor(engine: "Engine3D", which: int, comp: int, parity=None):
    """torch tensor aliasing one ghost-exchange plane of the engine (no copy)."""
    import torch
    ptr, nbytes = engine.halo(which, comp, parity)
    itemsize = 4 if engine.precision == 32 else 8
    return torch.as_tensor(_DeviceSpan(ptr, nbytes, itemsize), device=torch.device("cuda", torch.cuda.current_device()))
