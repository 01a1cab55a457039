#!/usr/bin/env python3
"""Is the step time independent of the data?  The headline workload (1024^3, eps/mu grids, 20-layer CPML,
fp32, fused storage: two steps per launch unless FDTD_PAIR=0) timed (a) from the cold start, where after 30 steps nearly every cell still holds zero,
(b) after the wave of the sheet source has crossed a good part of the grid, (c) with D and B filled with
noise at every cell.  The kernels have no data-dependent branch; this shows it.
    python tools/bench_data.py [edge]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fdtd_amd  # noqa: E402
from fdtd_amd import engine as E  # noqa: E402
from fdtd_amd import setup as fs  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = fs.constants()
d = (1053e-9 / 20,) * 3
dt = d[0] / (2 * K["c0"])
total = 2000
eng = fdtd_amd.Engine3D((n, n, n), (20,) * 6, d, dt, K["mu0"], 20, total + 64, precision=32, has_mu=True, mode=E.MODE_FUSED)
fs.load_cpml3d(eng, d, dt)
wf, mag = fs.default_source3d(n, n, total + 64, d, dt)
eng.set(E.JX_WF, wf)
eng.set(E.JX_MAG, mag)
eng.fill_materials(1234, K["eps0"], 4 * K["eps0"], K["mu0"], 2 * K["mu0"])


def timed(n0, count=20):
    torch.cuda.synchronize()
    eng.kernel_timing(True)
    t0 = time.perf_counter()
    eng.step(n0, count)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    eng.kernel_timing(False)
    n1, ms1 = eng.kernel_time(2)          # one step per launch
    n2, ms2 = eng.kernel_time(3)          # two steps per launch
    return el / count * 1e3, (ms1 + ms2) / max(n1 + 2 * n2, 1)


print(f"gaps GiB {[round(v, 2) for v in eng.spread_gaps()]}")
print("| state | steps done | wall ms/step | kernel ms/step | sum Dx^2 |\n|---|---|---|---|---|")
eng.step(0, 10)
done = 10
w, k = timed(done); done += 20
print(f"| cold start (source ramping up, wave a few cells from plane 20) | {done} | {w:.3f} | {k:.3f} | {eng.sumsq(E.DX):.3e} |", flush=True)
for stop in (500, 1000, 2000):
    eng.step(done, stop - done); done = stop
    w, k = timed(done); done += 20
    print(f"| wave front near plane {20 + int(done * 0.5 / 1.9)} of {n} | {done} | {w:.3f} | {k:.3f} | {eng.sumsq(E.DX):.3e} |", flush=True)
for id_ in (E.DX, E.DY, E.DZ):
    eng.fill_array(id_, 7, -1e-11, 1e-11)
for id_ in (E.BX, E.BY, E.BZ):
    eng.fill_array(id_, 7, -1e-8, 1e-8)
w, k = timed(done); done += 20
print(f"| D and B noise at every cell | {done} | {w:.3f} | {k:.3f} | {eng.sumsq(E.DX):.3e} |", flush=True)
