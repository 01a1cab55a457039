#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE COMPILED REFERENCE.

Run only where the reference checkout is mounted (it builds oracle/_ref from the reference's own
sources and drives the unmodified FDTD3D / FDTD2D classes through oracle/ref_harness_*.cpp):

    python tests/golden/make_golden.py

Every fixture is an .npz of float64 arrays in canonical reference order, holding BOTH the inputs
the step functions consume (CPML tables, eps, source magnitude/waveform, Lorentz coefficients --
so that checks on another machine do not depend on its libm) and the reference's outputs.
Keys: ``meta`` (json string), ``in_<id>`` inputs, ``out_s<step>_<id>`` outputs, ``probe`` series.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def seq_sumsq(a) -> float:
    """Left-to-right sum of squares, the order a plain C loop uses."""
    return float(np.cumsum(np.asarray(a, dtype=np.float64).ravel() ** 2)[-1])


def inputs3d(r, npml):
    ins = {}
    for a in range(3):
        for s in range(2):
            for k in range(6):
                ins[po.tab_id(a, s, k)] = r.get(po.tab_id(a, s, k))
    for id_ in (po.EPS_X, po.EPS_Y, po.EPS_Z, po.JX_MAG, po.JX_WF):
        ins[id_] = r.get(id_)
    return ins


def save(name, meta, ins, outs, extra=None):
    d = {"meta": np.array(json.dumps(meta))}
    for k, v in ins.items():
        d[f"in_{k}"] = v
    for (step, id_), v in outs.items():
        d[f"out_s{step}_{id_}"] = v
    d.update(extra or {})
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **d)
    print(f"{name}: {os.path.getsize(path) / 1e3:.0f} kB")


def g1():
    N, npml, steps, kJ = (16, 14, 24), (3,) * 6, 40, 12
    r = po.Ref3D(N, npml, steps)
    r.set_kjx(kJ)
    assert r.tables_consistent()
    s = r.scalars()
    ins = inputs3d(r, npml)
    outs = {}
    for n in range(steps):
        r.step(n)
        if n + 1 in (2, steps):
            for id_ in po.FIELDS3D:
                outs[(n + 1, id_)] = r.get(id_)
    for id_ in po.ALL_PHI3D:
        outs[(steps, id_)] = r.get(id_)
    meta = dict(kind="3d", N=N, npml=npml, steps=steps, k_Jx=kJ, snapshots=[2, steps],
                **{k: s[k] for k in ("dx", "dy", "dz", "dt", "mu0", "eps0")},
                note="CPML on six faces, reference default source and materials")
    save("g1_3d_cpml", meta, ins, outs)


def g2():
    N, npml, steps, kJ = (24, 20, 28), (0,) * 6, 120, 14
    r = po.Ref3D(N, npml, steps)
    r.set_kjx(kJ)
    s = r.scalars()
    for c in range(3):
        r.set(po.EPS_X + c, np.full(r.size(po.EPS_X + c), s["eps0"]))
    mag = np.zeros((N[0], N[1] + 1))
    mag[N[0] // 2, N[1] // 2] = 1.0
    r.set(po.JX_MAG, mag)
    ins = inputs3d(r, npml)
    probe_cell = (N[0] // 2, N[1] // 2 + 2, kJ + 3)
    probe = np.zeros(steps)
    outs = {}
    for n in range(steps):
        r.step(n)
        probe[n] = r.get(po.EX, po.shape3d(po.EX, N, npml))[probe_cell]
    for id_ in po.FIELDS3D:
        outs[(steps, id_)] = r.get(id_)
    meta = dict(kind="3d", N=N, npml=npml, steps=steps, k_Jx=kJ, snapshots=[steps], probe_cell=probe_cell,
                **{k: s[k] for k in ("dx", "dy", "dz", "dt", "mu0", "eps0")},
                note="vacuum cavity: no CPML (implicit PEC walls), eps = eps0, one-cell Jx dipole")
    save("g2_3d_cavity", meta, ins, outs, {"probe": probe})


def g3():
    N, npml, steps, kJ = (20, 18, 26), (3, 5, 2, 4, 6, 1), 30, 13
    r = po.Ref3D(N, npml, steps)
    r.set_kjx(kJ)
    s = r.scalars()
    seed = 1234
    for c in range(3):
        n = r.size(po.EPS_X + c)
        u = np.array([po.hash_unit(seed, c, i) for i in range(n)])
        r.set(po.EPS_X + c, s["eps0"] * 1.0 + (s["eps0"] * 4.0 - s["eps0"] * 1.0) * u)
    ins = inputs3d(r, npml)
    outs = {}
    r.step(0, steps)
    for id_ in po.FIELDS3D + po.ALL_PHI3D:
        outs[(steps, id_)] = r.get(id_)
    meta = dict(kind="3d", N=N, npml=npml, steps=steps, k_Jx=kJ, snapshots=[steps], seed=seed,
                eps_lo=s["eps0"] * 1.0, eps_hi=s["eps0"] * 4.0,
                **{k: s[k] for k in ("dx", "dy", "dz", "dt", "mu0", "eps0")},
                note="heterogeneous eps in [1,4] eps0 from the counter hash, asymmetric CPML")
    save("g3_3d_hetero", meta, ins, outs)


def inputs2d(r):
    ins = {}
    for a in range(2):
        for s in range(2):
            for k in range(6):
                id_ = po.D2_TAB + (a * 2 + s) * 6 + k
                ins[id_] = r.get(id_)
    for id_ in (po.D2_C1, po.D2_C2, po.D2_C3, po.D2_JY):
        ins[id_] = r.get(id_)
    return ins


def g4():
    Nx, Nz, npml, steps = 96, 112, (8,) * 4, 160
    r = po.Ref2D(Nx, Nz, npml, steps)
    s = r.scalars()
    ins = inputs2d(r)
    outs = {}
    r.step(0, steps)
    for id_ in po.FIELDS2D + po.ALL_PHI2D:
        outs[(steps, id_)] = r.get(id_)
    meta = dict(kind="2d", Nx=Nx, Nz=Nz, npml=npml, steps=steps, i_Jy=s["i_Jy"], k_Jy=s["k_Jy"], snapshots=[steps],
                **{k: s[k] for k in ("dx", "dz", "dt", "mu0", "eps0")},
                note="reference default Lorentz slab, CPML on four sides, default line source")
    save("g4_2d_slab", meta, ins, outs)


def g5():
    Nx, Nz, npml, steps = 512, 512, (0,) * 4, 2000
    r = po.Ref2D(Nx, Nz, npml, steps)
    for id_ in (po.D2_C1, po.D2_C2, po.D2_C3):
        r.set(id_, np.zeros(Nx * Nz))
    r.set_source_cell(Nx // 2, Nz // 2)
    s = r.scalars()
    ins = {po.D2_JY: r.get(po.D2_JY)}
    probe_cell = (Nx // 2 + 37, Nz // 2 - 21)
    probe = np.zeros(steps)
    outs, sums = {}, {}
    sh = po.shape2d(po.D2_EY, Nx, Nz, npml)
    for n in range(steps):
        r.step(n)
        ey = r.get(po.D2_EY, sh)
        probe[n] = ey[probe_cell]
        if (n + 1) % 500 == 0:
            outs[(n + 1, po.D2_EY)] = ey[::8, ::8].copy()
            sums[str(n + 1)] = seq_sumsq(ey)
    meta = dict(kind="2d", Nx=Nx, Nz=Nz, npml=npml, steps=steps, i_Jy=s["i_Jy"], k_Jy=s["k_Jy"],
                snapshots=[500, 1000, 1500, 2000], subsample=8, probe_cell=probe_cell, sumsq_ey=sums,
                **{k: s[k] for k in ("dx", "dz", "dt", "mu0", "eps0")},
                note="512x512, no CPML (PEC walls), no Lorentz material, point source at the centre; "
                     "Ey snapshots are every 8th cell in both directions")
    save("g5_2d_pec512", meta, ins, outs, {"probe": probe})


def _pack_dir(src, dst):
    import gzip
    os.makedirs(dst, exist_ok=True)
    for f in os.listdir(dst):
        os.remove(os.path.join(dst, f))
    for f in sorted(os.listdir(src)):
        with open(os.path.join(src, f), "rb") as fi, gzip.GzipFile(os.path.join(dst, f + ".gz"), "wb", mtime=0) as fo:
            fo.write(fi.read())
    print(dst, sorted(os.listdir(dst)))


def g7():
    """Output files of the reference's own run() on a small 3-D grid (VTI, ASCII)."""
    N, npml, steps = (16, 14, 24), (3,) * 6, 41
    r = po.Ref3D(N, npml, steps)
    s = r.scalars()
    out = r.run()
    _pack_dir(out, os.path.join(OUT, "run3d"))
    with open(os.path.join(OUT, "run3d", "meta.json"), "w") as f:
        json.dump(dict(N=N, npml=npml, steps=steps, k_Jx=s["k_Jx"], note="reference run(): default source and materials, "
                       "frames every 20 steps, RMS window of 20 steps; files gzip-compressed verbatim"), f, indent=1)


def g8():
    """Output files of the reference's own run() on a small 2-D grid (CSV)."""
    Nx, Nz, npml, steps = 48, 40, (5,) * 4, 81
    r = po.Ref2D(Nx, Nz, npml, steps)
    r.set_source_cell(12, 20)
    out = r.run()
    _pack_dir(out, os.path.join(OUT, "run2d"))
    with open(os.path.join(OUT, "run2d", "meta.json"), "w") as f:
        json.dump(dict(Nx=Nx, Nz=Nz, npml=npml, steps=steps, i_Jy=12, k_Jy=20, note="reference run(): default Lorentz slab and "
                       "source waveform, frames every 20 steps + final, RMS window of 40 steps; files gzip-compressed verbatim"), f, indent=1)


def g6():
    r2 = po.Ref2D()
    r2.step(0, 200)
    r3 = po.Ref3D()
    r3.step(0, 5)
    ka = dict(sumsq_ey_2d_default_200_steps=seq_sumsq(r2.get(po.D2_EY)),
              sumsq_ex_3d_default_5_steps=seq_sumsq(r3.get(po.EX)),
              note="unmodified reference defaults (200x480, 1580 steps; 200^3, 1501 steps); "
                   "left-to-right double sums over the arrays in reference order")
    with open(os.path.join(OUT, "g6_known_answers.json"), "w") as f:
        json.dump(ka, f, indent=1)
    print(ka)


def checksum(a) -> int:
    """The engine's order-independent checksum (fdtd3d_checksum) of a float64 array in canonical order:
    value bits mixed with the flat canonical index, summed modulo 2^64 over the non-zero entries."""
    v = np.ascontiguousarray(a, dtype=np.float64).ravel()
    idx = np.flatnonzero(v != 0.0).astype(np.uint64)
    bits = v[idx.astype(np.int64)].view(np.uint64)
    with np.errstate(over="ignore"):
        z = bits + np.uint64(0x9E3779B97F4A7C15) * (idx + np.uint64(0x632BE59BD9B4E019))
        for _ in range(2):
            z = (z ^ (z >> np.uint64(32))) * np.uint64(0xD6E8FEB86659FD93)
        z = z ^ (z >> np.uint64(32))
        return int(z.sum(dtype=np.uint64))


def g9():
    """The UNMODIFIED 3-D default (200^3, 20-layer CPML, glass with an air sphere, Gaussian sheet source)
    over its full run length of 1501 steps (FDTD3D.cpp:39): probe series at six cells for every step;
    at steps 200, 750 and 1501 every 8th cell of all E and H components, the left-to-right sums of
    squares and the engine-style checksums of all twelve field arrays."""
    r = po.Ref3D()
    s = r.scalars()
    N, steps, npml = r.N, s["steps"], (20,) * 6
    assert N == (200, 200, 200) and steps == 1501
    ins = {}
    for a in range(3):
        for sd in range(2):
            for k in range(6):
                ins[po.tab_id(a, sd, k)] = r.get(po.tab_id(a, sd, k))
    ins[po.JX_MAG] = r.get(po.JX_MAG)
    ins[po.JX_WF] = r.get(po.JX_WF)
    # eps itself is regenerated on the device; the engine does not store the wall column i = Nx of the components
    # with Nx+1 columns (no sweep reads it), so a second set of checksums leaves that column out
    eps_sums, eps_sums_stored = {}, {}
    for c in range(3):
        a = r.get(po.EPS_X + c, po.shape3d(po.EPS_X + c, N, npml)).copy()
        eps_sums[str(po.EPS_X + c)] = checksum(a)
        if a.shape[0] == N[0] + 1:
            a[N[0]] = 0.0
        eps_sums_stored[str(po.EPS_X + c)] = checksum(a)
    probes = [(po.EX, (100, 100, 30)), (po.EX, (100, 100, 100)), (po.EY, (120, 80, 140)), (po.EZ, (90, 110, 60)),
              (po.HY, (100, 100, 170)), (po.EX, (10, 190, 185))]
    flat = []
    for id_, cell in probes:
        sh = po.shape3d(id_, N, npml)
        flat.append(int(np.ravel_multi_index(cell, sh)))
    series = np.zeros((steps, len(probes)))
    snaps = (200, 750, 1501)
    outs, sums, cks = {}, {}, {}
    ids = (po.EX, po.EY, po.EZ, po.HX, po.HY, po.HZ)
    import time
    t0 = time.time()
    for n in range(steps):
        r.step(n)
        for q, (id_, _) in enumerate(probes):
            series[n, q] = r.peek(id_, flat[q])
        if n + 1 in snaps:
            for id_ in po.FIELDS3D:
                a = r.get(id_, po.shape3d(id_, N, npml))
                sums[f"{n + 1}:{id_}"] = seq_sumsq(a)
                cks[f"{n + 1}:{id_}"] = checksum(a)
                if id_ in ids:
                    outs[(n + 1, id_)] = a[::8, ::8, ::8].copy()
            print(f"g9: step {n + 1} after {time.time() - t0:.0f} s", flush=True)
    meta = dict(kind="3d", N=N, npml=npml, steps=steps, k_Jx=s["k_Jx"], snapshots=list(snaps), subsample=8,
                probes=[[id_, list(cell)] for id_, cell in probes], sumsq=sums, checksums=cks, eps_checksums=eps_sums,
                eps_checksums_without_wall_column=eps_sums_stored,
                **{k: s[k] for k in ("dx", "dy", "dz", "dt", "mu0", "eps0")},
                note="unmodified reference default problem over its full 1501 steps; eps is not stored (the engine "
                     "regenerates it with fdtd3d_fill_reference_materials; its checksums are)")
    save("g9_3d_default_full_run", meta, ins, outs, {"probe": series})


def g10():
    """The UNMODIFIED 2-D default (200x480, Lorentz slab, 20-layer CPML) over its full 1580 steps
    (FDTD2D.cpp:38): probe series at five cells for every step; Ey, Hx, Hz, SEy in full at steps 400,
    1000 and 1580."""
    r = po.Ref2D()
    s = r.scalars()
    Nx, Nz, steps, npml = s["Nx"], s["Nz"], s["steps"], (20,) * 4
    assert (Nx, Nz, steps) == (200, 480, 1580)
    ins = inputs2d(r)
    probes = [(po.D2_EY, (60, 240)), (po.D2_EY, (100, 240)), (po.D2_EY, (140, 240)), (po.D2_HX, (160, 240)), (po.D2_EY, (100, 30))]
    flat = [int(np.ravel_multi_index(cell, po.shape2d(id_, Nx, Nz, npml))) for id_, cell in probes]
    series = np.zeros((steps, len(probes)))
    snaps = (400, 1000, 1580)
    outs, sums = {}, {}
    for n in range(steps):
        r.step(n)
        for q, (id_, _) in enumerate(probes):
            series[n, q] = r.peek(id_, flat[q])
        if n + 1 in snaps:
            for id_ in (po.D2_EY, po.D2_HX, po.D2_HZ, po.D2_SEY, po.D2_DY):
                a = r.get(id_)
                outs[(n + 1, id_)] = a
                sums[f"{n + 1}:{id_}"] = seq_sumsq(a)
    meta = dict(kind="2d", Nx=Nx, Nz=Nz, npml=npml, steps=steps, i_Jy=s["i_Jy"], k_Jy=s["k_Jy"], snapshots=list(snaps),
                probes=[[id_, list(cell)] for id_, cell in probes], sumsq=sums,
                **{k: s[k] for k in ("dx", "dz", "dt", "mu0", "eps0")},
                note="unmodified reference default problem over its full 1580 steps")
    save("g10_2d_default_full_run", meta, ins, outs, {"probe": series})


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10"]
    for w in which:
        globals()[w]()
