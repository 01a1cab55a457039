#!/usr/bin/env python3
"""fdtd2d measurement line (BASELINE config 2): 8192 x 8192 cells, 20-layer CPML on four sides, fp32, on
one MI355X.  Not the graded headline (bench.py is); prints ONE JSON line with the same roofline fields.

    python tools/bench2d.py [--material slab|everywhere] [--mode fused|full] [--n 8192] [--steps 100]

--material slab        the reference's default material: a Lorentz slab 41 columns wide (FDTD2D.cpp:84-112),
                       every other cell plain (C1 = C2 = C3 = 0)
--material everywhere  the same Lorentz coefficients at every cell: the general path of FDTD2D::stepOnce

Algorithmic bytes per cell and step (fp32), by cell class:
  dispersive cell  33 words = 132 B: the reference's stepOnce with its histories rotated by pointer instead of
                   copied (E part 13 words, H part 20 words; FDTD2D.cpp:365-379, :456-481)
  plain cell        6 words =  24 B: Dy, Bx, Bz read and written once; with C = 0 the recursion reduces to
                   Ey = Dy/eps0, H = B/mu0 bit for bit, so no field or history sample has to exist in memory
roofline.achieved = (cells of each class x its bytes) / device time of the step's kernels, from HIP events on
the launch stream."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import fdtd_amd  # noqa: E402
from fdtd_amd import engine as E  # noqa: E402
from fdtd_amd import setup as fs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--material", default="slab", choices=["slab", "everywhere"])
ap.add_argument("--mode", default="fused", choices=["fused", "full"])
ap.add_argument("--n", type=int, default=int(os.environ.get("FDTD_BENCH2D_N", "8192")))
ap.add_argument("--steps", type=int, default=int(os.environ.get("STEPS", "100")))
ap.add_argument("--warmup", type=int, default=10)
args = ap.parse_args()

N, steps, warm = args.n, args.steps, args.warmup
K = fs.constants()
dx = K["c0"] / 1.5e14 / 40
dt = dx / (2 * K["c0"])
eng = fdtd_amd.Engine2D(N, N, (20,) * 4, dx, dx, dt, K["mu0"], K["eps0"], N // 2 - 40, N // 2, steps + warm, precision=32,
                        mode=E.MODE2D_FUSED if args.mode == "fused" else E.MODE2D_FULL)
fs.load_cpml2d(eng, dx, dx, dt)
eng.set(E.D2_JY, fs.default_source2d(steps + warm, dx, dx, dt))
coef = fs.default_lorentz2d(N, N, dt)
if args.material == "everywhere":
    inside = coef[0][N // 2, 0]
    assert inside != 0
    coef = [np.full((N, N), c[N // 2, 0]) for c in coef]
n_disp = int(np.count_nonzero((coef[0] != 0) | (coef[1] != 0) | (coef[2] != 0)))
for q, c in enumerate(coef):
    eng.set(E.D2_C1 + q, c)
del coef
eng.step(0, warm)
torch.cuda.synchronize()
eng.kernel_timing(True)
t0 = time.perf_counter()
eng.step(warm, steps)
torch.cuda.synchronize()
el = time.perf_counter() - t0
ne, me = eng.kernel_time(0)
nh, mh = eng.kernel_time(1)
nf, mf = eng.kernel_time(2)
t_kernels = (me + mh + mf) / steps * 1e-3                 # device seconds per step
cells = N * N
alg_bytes = n_disp * 132 + (cells - n_disp) * 24
ach = alg_bytes / t_kernels / 1e9
print(json.dumps({
    "metric": "Mcell-updates/s", "value": cells * steps / el / 1e6, "unit": "Mcell-updates/s", "n_gpus": 1, "steps": steps, "warmup": warm,
    "ms_per_step": el / steps * 1e3, "higher_is_better": True, "dtype": "f32", "data": "synthetic",
    "config": {"workload": f"fdtd2d {N}x{N}, 20-layer CPML on four sides, "
                           + ("reference Lorentz slab (41 columns), plain elsewhere" if args.material == "slab" else "Lorentz-dispersive at every cell"),
               "stepping": "one kernel per step (k2_fused)" if args.mode == "fused" else "two kernels per step (k2_part_e, k2_part_h), every array resident",
               "dispersive_cells": n_disp, "plain_cells": cells - n_disp, "device_bytes": eng.device_bytes()},
    "roofline": {"bound": "hbm", "kernel": "k2_fused" if args.mode == "fused" else "k2_part_e+k2_part_h", "achieved": ach, "peak": 8000.0,
                 "unit": "GB/s", "frac": ach / 8000.0, "traffic": None,
                 "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_bytes_per_dispersive_cell": 132, "algorithmic_bytes_per_plain_cell": 24,
                 "reference_formulation_everywhere_frac": cells * 132 / t_kernels / 1e9 / 8000.0,
                 "kernel_ms_per_step": t_kernels * 1e3, "ms_e_part": me / max(ne, 1), "ms_h_part": mh / max(nh, 1), "ms_fused": mf / max(nf, 1)},
    "checks": {"sumsq_ey": eng.sumsq(E.D2_EY), "finite": bool(np.isfinite(eng.sumsq(E.D2_EY)))}}))
