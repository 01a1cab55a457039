#!/usr/bin/env python3
"""2-D stepping with the RMS window of Ey advanced after every step (the body of the reference's time loop), 8192^2 with the
reference's Lorentz slab: fdtd2d_step_observed (two steps per launch, the window advanced inside the launches) against
one fdtd2d_step + fdtd2d_observe_accumulate per step.   python tools/bench2d_observed.py [--n 8192] [--steps 200]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import fdtd_amd
from fdtd_amd import engine as E, setup as fs

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=8192)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--window", type=int, default=40)
a = ap.parse_args()
K = fs.constants()
dx = K["c0"] / 1.5e14 / 40
dt = dx / (2 * K["c0"])
out = {"grid": [a.n, a.n], "steps": a.steps, "window": a.window}
for kind in ("observed", "step_by_step"):
    eng = fdtd_amd.Engine2D(a.n, a.n, (20,) * 4, dx, dx, dt, K["mu0"], K["eps0"], max(a.n // 2 - 40, 1), a.n // 2, a.steps + 20, precision=32, mode=E.MODE2D_FUSED)
    fs.load_cpml2d(eng, dx, dx, dt)
    eng.set(E.D2_JY, fs.default_source2d(a.steps + 20, dx, dx, dt))
    for q, c in enumerate(fs.default_lorentz2d(a.n, a.n, dt)):
        eng.set(E.D2_C1 + q, c)
    w = 4.0 / (2 * a.window)
    run = (lambda n0, cnt: eng.step_observed(n0, cnt, w, a.window)) if kind == "observed" else \
          (lambda n0, cnt: [(eng.step(n, 1), eng.observe_accumulate(w, n % a.window == 0)) for n in range(n0, n0 + cnt)])
    run(0, 20); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(20, a.steps); torch.cuda.synchronize()
    out[kind] = {"ms_per_step": (time.perf_counter() - t0) / a.steps * 1e3, "sumsq_window": float(np.sum(eng.observe_rms() ** 2))}
    eng.close()
print(json.dumps(out))
