#!/usr/bin/env python3
"""A few time steps of the 3-D engine at a given size and mode, for use under rocprofv3.

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -- python3 tools/prof3d.py --mode fused --n 1024 --steps 4
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import fdtd_amd  # noqa: E402
from fdtd_amd import setup as fs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="fused", choices=["full", "lean", "fused"])
ap.add_argument("--n", type=int, default=1024)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--chunk", type=int, default=0)
ap.add_argument("--rows", type=int, default=0)
ap.add_argument("--precision", type=int, default=32)
ap.add_argument("--no-mu", action="store_true")
ap.add_argument("--npml", type=int, default=20)
ap.add_argument("--noise", action="store_true", help="every cell of D and B non-zero before the steps (counter-hash noise)")
args = ap.parse_args()

n = args.n
K = fs.constants()
d = (1053e-9 / 20,) * 3
dt = d[0] / (K["c0"] * 2)
eng = fdtd_amd.Engine3D((n, n, n), (args.npml,) * 6, d, dt, K["mu0"], 20, 4096, precision=args.precision, has_mu=not args.no_mu,
                        mode={"full": 0, "lean": 1, "fused": 2}[args.mode])
if args.npml:
    fs.load_cpml3d(eng, d, dt)
wf, mag = fs.default_source3d(n, n, 4096, d, dt)
eng.set(fdtd_amd.engine.JX_WF, wf)
eng.set(fdtd_amd.engine.JX_MAG, mag)
eng.fill_materials(1234, K["eps0"], 4 * K["eps0"], K["mu0"], 2 * K["mu0"])
if args.chunk or args.rows:
    eng.tune(args.chunk, -1, args.rows)
if os.environ.get("FDTD_SHAPE_OUT"):            # for tools/pmc_traffic.py: the launch shape these launches use
    import json
    with open(os.environ["FDTD_SHAPE_OUT"], "w") as f:
        json.dump(eng.stamp_shape(), f)
if args.noise:
    for c in range(3):
        eng.fill_array(fdtd_amd.engine.DX + c, 7, -1e-11, 1e-11)
        eng.fill_array(fdtd_amd.engine.BX + c, 7, -1e-8, 1e-8)
eng.step(0, args.steps)
torch.cuda.synchronize()
print("done", args.mode, n, args.steps)
