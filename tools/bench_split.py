#!/usr/bin/env python3
"""What the split step of a slab costs on its own GPU, without any exchange:
    python tools/bench_split.py [edge] [layers]
One slab engine holding layers [layers, 2*layers) of an edge x edge x 3*layers problem (a middle slab:
neighbours on both sides) is stepped (a) with one launch per step (PART_ALL + the top-layer kernel) and
(b) the way a slab stepper does it: EDGE, TOP, top-layer kernel, INTERIOR.  The ghost planes keep their
initial zeros; only the launch structure is measured."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd_amd  # noqa: E402
from fdtd_amd import engine as E  # noqa: E402
from fdtd_amd import setup as fs  # noqa: E402

edge = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
layers = int(sys.argv[2]) if len(sys.argv) > 2 else edge
K = fs.constants()
d = (1053e-9 / 20,) * 3
dt = d[0] / (2 * K["c0"])
warm, steps = 6, 20
eng = fdtd_amd.Engine3D((edge, edge, 3 * layers), (20,) * 6, d, dt, K["mu0"], layers + 20, 2 * (warm + steps), precision=32,
                        has_mu=True, mode=E.MODE_FUSED, z_range=(layers, 2 * layers))
fs.load_cpml3d(eng, d, dt)
wf, mag = fs.default_source3d(edge, edge, 2 * (warm + steps), d, dt)
eng.set(E.JX_WF, wf)
eng.set(E.JX_MAG, mag)
eng.fill_materials(1234, K["eps0"], 4 * K["eps0"], K["mu0"], 2 * K["mu0"])
L = E.lib()


def sync():
    import ctypes as C
    v = C.c_double()
    L.fdtd3d_sumsq(eng.h, C.c_int(E.DX), C.byref(v), None)      # synchronises the null stream


def one(n):
    eng.fused(n, E.PART_ALL)
    eng.fused_top()


def split(n):
    eng.fused(n, E.PART_EDGE)
    eng.fused(n, E.PART_TOP)
    eng.fused_top()
    eng.fused(n, E.PART_INTERIOR)


out = {}
n = 0
for name, fn in (("one launch", one), ("split", split), ("one launch", one), ("split", split)):
    for _ in range(warm):
        fn(n); n += 1
    sync()
    eng.kernel_timing(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn(n); n += 1
    sync()
    el = time.perf_counter() - t0
    eng.kernel_timing(False)
    nf, msf = eng.kernel_time(2)
    nh, msh = eng.kernel_time(1)
    print(f"{name:10s}: {el / steps * 1e3:7.3f} ms/step wall; fused launches {nf / steps:.0f}/step {msf / steps:7.3f} ms, "
          f"top-layer kernel {msh / steps:6.3f} ms; gaps GiB {[round(v, 2) for v in eng.spread_gaps()]}", flush=True)
