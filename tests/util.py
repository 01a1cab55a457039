"""Shared helpers for the parity tests: fixture loading and back-end-agnostic drivers.

A "back end" is anything with ``set(id, array)``, ``get(id) -> flat float64 array`` and
``step(n0, count)`` over the array ids of include/fdtd_engine.h: the CPU twin, the compiled
reference, or the HIP engine wrapper.
"""
import json
import os

import numpy as np

from oracle import pyoracle as po

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_fixture(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    ins, outs, extra = {}, {}, {}
    for k in z.files:
        if k.startswith("in_"):
            ins[int(k[3:])] = z[k]
        elif k.startswith("out_s"):
            step, id_ = k[5:].split("_")
            outs[(int(step), int(id_))] = z[k]
        elif k != "meta":
            extra[k] = z[k]
    return meta, ins, outs, extra


def twin3d_from_meta(meta, prec, has_mu=False):
    return po.Twin3D(prec, meta["N"], meta["npml"], (meta["dx"], meta["dy"], meta["dz"]), meta["dt"],
                     meta["mu0"], meta["k_Jx"], meta["steps"], has_mu)


def twin2d_from_meta(meta, prec):
    return po.Twin2D(prec, meta["Nx"], meta["Nz"], meta["npml"], meta["dx"], meta["dz"], meta["dt"],
                     meta["mu0"], meta["eps0"], meta["i_Jy"], meta["k_Jy"], meta["steps"])


def feed(backend, ins):
    for id_, arr in ins.items():
        if arr.size:
            backend.set(id_, arr)


def run_snapshots(backend, meta, outs, on_step=None):
    """Advance to every snapshot step of the fixture; return {(step,id): array} for the ids in outs."""
    got, n = {}, 0
    for snap in sorted(meta["snapshots"]):
        if on_step is None:
            backend.step(n, snap - n)
        else:
            for m in range(n, snap):
                backend.step(m, 1)
                on_step(m)
        n = snap
        for (step, id_) in outs:
            if step == snap:
                got[(step, id_)] = backend.get(id_)
    return got


def assert_equal_arrays(got, want, what, sub=None):
    """Bit-level equality of values (+0 == -0 allowed, NaN never expected)."""
    g = np.asarray(got, dtype=np.float64).reshape(-1)
    w = np.asarray(want, dtype=np.float64).reshape(-1)
    assert g.shape == w.shape, f"{what}: shape {g.shape} vs {w.shape}"
    assert np.all(np.isfinite(g)), f"{what}: non-finite values"
    if not np.array_equal(g, w):
        bad = np.flatnonzero(g != w)
        i = bad[0]
        scale = np.abs(w).max()
        raise AssertionError(f"{what}: {bad.size}/{g.size} values differ, first at {i}: {g[i]!r} vs {w[i]!r}; "
                             f"max |diff| {np.abs(g - w).max():.3e}, max |ref| {scale:.3e}")


def rel_l2(got, want) -> float:
    g = np.asarray(got, dtype=np.float64).ravel()
    w = np.asarray(want, dtype=np.float64).ravel()
    den = np.linalg.norm(w)
    return float(np.linalg.norm(g - w) / den) if den > 0 else float(np.linalg.norm(g))


def family_rel_l2(got_list, want_list) -> float:
    """Relative L2 error of a vector field given as a list of component arrays."""
    num = sum(float(np.sum((np.asarray(g, dtype=np.float64).ravel() - np.asarray(w, dtype=np.float64).ravel()) ** 2))
              for g, w in zip(got_list, want_list))
    den = sum(float(np.sum(np.asarray(w, dtype=np.float64) ** 2)) for w in want_list)
    return float(np.sqrt(num / den)) if den > 0 else float(np.sqrt(num))


def seq_sumsq(a) -> float:
    return float(np.cumsum(np.asarray(a, dtype=np.float64).ravel() ** 2)[-1])
