#!/usr/bin/env python3
"""Step times of the other BASELINE configurations and storage modes on one MI355X (markdown table rows):
    python tools/bench_configs.py"""
import os, sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch, numpy as np
import fdtd_amd
from fdtd_amd import setup as fs, engine as E
print("| configuration | ms/step | Gcell/s | GiB between arrays |\n|---|---|---|---|")
K = fs.constants(); d = (1053e-9/20,)*3; dt = d[0]/(K["c0"]*2)
def run(tag, N, npml, prec, has_mu, mode, dipole=False, steps=20):
    eng = fdtd_amd.Engine3D(N, npml, d, dt, K["mu0"], N[2]//2 if dipole else 20, 4096, precision=prec, has_mu=has_mu, mode=mode)
    if any(npml): fs.load_cpml3d(eng, d, dt)
    wf, mag = fs.default_source3d(N[0], N[1], 4096, d, dt)
    if dipole:
        mag[:] = 0; mag[N[0]//2, N[1]//2] = 1.0
    eng.set(E.JX_WF, wf); eng.set(E.JX_MAG, mag)
    if has_mu: eng.fill_materials(1234, K["eps0"], 4*K["eps0"], K["mu0"], 2*K["mu0"])
    elif dipole: eng.fill_materials(1234, K["eps0"], K["eps0"], K["mu0"], K["mu0"])
    else: eng.fill_materials(1234, K["eps0"], 4*K["eps0"], K["mu0"], K["mu0"])
    if os.environ.get("FDTD_TUNE_ROWS"):          # experiment: rows per workgroup of the fused kernel
        eng.tune(int(os.environ.get("FDTD_TUNE_CHUNK", "0")), -1, int(os.environ["FDTD_TUNE_ROWS"]))
    eng.step(0, 5); torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.step(5, steps); torch.cuda.synchronize(); el = (time.perf_counter()-t0)/steps
    cells = N[0]*N[1]*N[2]
    print(f"| {tag} | {el*1e3:.2f} | {cells/el/1e9:.1f} | {eng.spread_gap_gib():.1f} |", flush=True)
    eng.close()
run("512^3 vacuum cavity, dipole, no CPML, fp32 fused", (512,)*3, (0,)*6, 32, False, 2, dipole=True, steps=100)
run("1024^3 eps grid, constant mu0 (reference formulation, 108 B/cell), CPML, fp32 fused", (1024,)*3, (20,)*6, 32, False, 2)
run("1024^3 eps+mu grids, CPML, fp32 fused", (1024,)*3, (20,)*6, 32, True, 2)
run("1024^3 eps+mu grids, CPML, fp32 lean", (1024,)*3, (20,)*6, 32, True, 1)
run("1024^3 eps+mu grids, CPML, fp32 full", (1024,)*3, (20,)*6, 32, True, 0)
run("768^3 eps+mu grids, CPML, fp64 fused", (768,)*3, (20,)*6, 64, True, 2)
run("1024x1024x3072 eps+mu grids, CPML, fp32 fused (242 GB)", (1024,1024,3072), (20,)*6, 32, True, 2, steps=6)
