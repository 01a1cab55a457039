#!/usr/bin/env python3
"""Step time of a slab group (one host thread, streams, device-to-device ghost copies):
    python tools/bench_group.py [n_slabs] [edge] [mode]
On a one-GPU box all slabs share the device, so this measures what the choreography costs; on a
multi-GPU node slab r runs on GPU r and the number is the whole-node rate for edge x edge x (edge*n_slabs)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fdtd_amd import engine as E  # noqa: E402
from fdtd_amd import setup as fs  # noqa: E402

n_slabs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
edge = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
mode = {"fused": E.MODE_FUSED, "lean": E.MODE_LEAN, "full": E.MODE_FULL}[sys.argv[3] if len(sys.argv) > 3 else "fused"]
ndev = E.lib().fdtd_device_count()
weak = ndev >= n_slabs                     # enough GPUs: one full-size slab each
Nz = edge * n_slabs if weak else edge
K = fs.constants()
d = (1053e-9 / 20,) * 3
dt = d[0] / (2 * K["c0"])
warm, steps = 6, 20
g = E.EngineGroup3D((edge, edge, Nz), (20,) * 6, d, dt, K["mu0"], 20, warm + steps, n_slabs, precision=32, has_mu=True, mode=mode)
wf, mag = fs.default_source3d(edge, edge, warm + steps, d, dt)
for m in g.members:
    fs.load_cpml3d(m, d, dt)
    m.set(E.JX_WF, wf)
    m.set(E.JX_MAG, mag)
g.fill_materials(1234, K["eps0"], 4 * K["eps0"], K["mu0"], 2 * K["mu0"])
g.step(0, warm)
g.sync()
t0 = time.perf_counter()
g.step(warm, steps)
g.sync()
el = time.perf_counter() - t0
print(f"{n_slabs} slab(s) on {min(ndev, n_slabs)} GPU(s), {edge}x{edge}x{Nz}, mode {mode}, {'two steps' if g.pairs else 'one step'} per launch: {el / steps * 1e3:.3f} ms/step, "
      f"{edge * edge * Nz * steps / el / 1e9:.2f} Gcell/s, checksum Dx {g.checksum(E.DX)}, "
      f"gaps GiB {[ [round(v, 2) for v in m.spread_gaps()] for m in g.members]}", flush=True)
