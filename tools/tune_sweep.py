#!/usr/bin/env python3
"""Launch-shape sweep of the fused kernel on one engine (the headline workload): planes per workgroup, tile order,
rows per workgroup.  python tools/tune_sweep.py [edge]"""
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
eng = fdtd_amd.Engine3D((n, n, n), (20,) * 6, d, dt, K["mu0"], 20, 4096, precision=32, has_mu=True, mode=E.MODE_FUSED)
fs.load_cpml3d(eng, d, dt)
wf, mag = fs.default_source3d(n, n, 4096, d, dt)
eng.set(E.JX_WF, wf)
eng.set(E.JX_MAG, mag)
eng.fill_materials(1234, K["eps0"], 4 * K["eps0"], K["mu0"], 2 * K["mu0"])
print("gap GiB", eng.spread_gap_gib(), "gaps", eng.spread_gaps(), flush=True)
step = [0]


def timed(count=6):
    eng.step(step[0], 2); step[0] += 2
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.step(step[0], count); step[0] += count
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / count * 1e3


base = timed()
print(f"default {eng.launch_shape()}: {base:.3f} ms", flush=True)
for rows in (16, 8):
    for chunk in (16, 24, 32, 48):
        for remap in (2, 4, 8):
            eng.tune(chunk, remap, rows)
            print(f"rows {rows:2d} chunk {chunk:2d} remap {remap:2d}: {timed():.3f} ms", flush=True)
