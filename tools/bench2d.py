#!/usr/bin/env python3
"""Secondary benchmark: fdtd2d 8192 x 8192, 20-layer CPML on four sides, the reference's Lorentz slab
on one MI355X.  Prints one JSON line; not the graded headline (bench.py is)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import fdtd_amd
from fdtd_amd import setup as fs
from fdtd_amd import engine as E

N = int(os.environ.get("FDTD_BENCH2D_N", "8192"))
steps, warm = int(os.environ.get("STEPS", "100")), 10
K = fs.constants()
dx = K["c0"] / 1.5e14 / 40
dt = dx / (2 * K["c0"])
MODE = os.environ.get("FDTD_BENCH2D_MODE", "fused")          # fused: one kernel per step; full: two kernels, every array resident
eng = fdtd_amd.Engine2D(N, N, (20,) * 4, dx, dx, dt, K["mu0"], K["eps0"], N // 2 - 40, N // 2, steps + warm, precision=32,
                        mode=E.MODE2D_FUSED if MODE == "fused" else E.MODE2D_FULL)
fs.load_cpml2d(eng, dx, dx, dt)
eng.set(E.D2_JY, fs.default_source2d(steps + warm, dx, dx, dt))
for q, c in enumerate(fs.default_lorentz2d(N, N, dt)):
    eng.set(E.D2_C1 + q, c)
eng.step(0, warm); torch.cuda.synchronize()
eng.kernel_timing(True)
t0 = time.perf_counter(); eng.step(warm, steps); torch.cuda.synchronize(); el = time.perf_counter() - t0
ne, me = eng.kernel_time(0); nh, mh = eng.kernel_time(1); nf, mf = eng.kernel_time(2)
cells = N * N
print(json.dumps({"metric": "Mcell-updates/s", "value": cells * steps / el / 1e6, "ms_per_step": el / steps * 1e3,
                  "config": {"workload": f"fdtd2d {N}x{N}, 20-layer CPML, Lorentz slab, fp32", "mode": MODE},
                  "roofline": {"bound": "hbm", "algorithmic_bytes_per_cell": 132, "achieved": cells * steps * 132 / el / 1e9, "peak": 8000.0,
                               "frac": cells * steps * 132 / el / 1e9 / 8000.0, "ms_e_part": me / max(ne, 1), "ms_h_part": mh / max(nh, 1), "ms_fused": mf / max(nf, 1)},
                  "checks": {"sumsq_ey": eng.sumsq(E.D2_EY)}}))
