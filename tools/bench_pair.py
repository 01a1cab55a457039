#!/usr/bin/env python3
"""One step per launch (k3_fused) against two steps per launch (k3_pair) on a whole-grid engine:
    python tools/bench_pair.py [--n 1024] [--npml 0] [--mu 1] [--prec 32] [--steps 20] [--chunk 48] [--rows 16] [--only pair|single]
Prints one JSON line; --check compares checksums of D and B after the same number of steps."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fdtd_amd
from fdtd_amd import setup as fs, engine as E

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1024)
ap.add_argument("--nz", type=int, default=0)
ap.add_argument("--ny", type=int, default=0)
ap.add_argument("--nx", type=int, default=0)
ap.add_argument("--mu", type=int, default=1)
ap.add_argument("--prec", type=int, default=32)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--chunk", type=int, default=0)
ap.add_argument("--rows", type=int, default=0)
ap.add_argument("--only", default="")
ap.add_argument("--remap", type=int, default=-1)
ap.add_argument("--split", type=int, default=0, help="two-step launches as edge planes + bulk (what the slab stepper does)")
ap.add_argument("--check", type=int, default=1)
ap.add_argument("--probes", type=int, default=0, help="register this many E probes, each on a plane of its own")
ap.add_argument("--npml", default="0", help="layers on every face, or six numbers x-,x+,y-,y+,z-,z+")
ap.add_argument("--noise", type=int, default=0, help="time a state with every cell of D and B non-zero (counter-hash noise)")
a = ap.parse_args()
a.npml = tuple(int(v) for v in a.npml.split(",")) if "," in a.npml else (int(a.npml),) * 6
K = fs.constants(); d = (1053e-9 / 20,) * 3; dt = d[0] / (K["c0"] * 2)
N = (a.nx or a.n, a.ny or a.n, a.nz or a.n)
cells = N[0] * N[1] * N[2]
out = {"grid": list(N), "npml": list(a.npml), "precision": a.prec, "mu_grid": bool(a.mu), "steps": a.steps, "probes": a.probes}


def make():
    eng = fdtd_amd.Engine3D(N, a.npml, d, dt, K["mu0"], N[2] // 2, 4096, precision=a.prec, has_mu=bool(a.mu), mode=E.MODE_FUSED)
    if any(a.npml):
        fs.load_cpml3d(eng, d, dt)
    wf, mag = fs.default_source3d(N[0], N[1], 4096, d, dt)
    eng.set(E.JX_WF, wf); eng.set(E.JX_MAG, mag)
    eng.fill_materials(1234, K["eps0"], 4 * K["eps0"], K["mu0"], (2 if a.mu else 1) * K["mu0"])
    if a.noise:
        for c in range(3):
            eng.fill_array(E.DX + c, 7, -1e-11, 1e-11)
            eng.fill_array(E.BX + c, 7, -1e-8, 1e-8)
    for q in range(a.probes):
        eng.probe_add(E.EX + q % 3, N[0] // 2, N[1] // 3, (q + 1) * N[2] // (a.probes + 1))
    return eng


def timed(eng, fn):
    fn(0, 3); torch.cuda.synchronize()          # warm-up: 6 steps either way
    t0 = time.perf_counter(); fn(6, a.steps // 2); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (a.steps // 2 * 2)


sums = {}
for kind in ("single", "pair"):
    if a.only and a.only != kind:
        continue
    if kind == "single":
        os.environ["FDTD_PAIR"] = "0"          # step() would otherwise go two steps per launch by itself on a large grid
    else:
        os.environ.pop("FDTD_PAIR", None)
    eng = make()
    if kind == "pair":
        eng.pair_shape(a.chunk, a.rows)
        if a.remap >= 0:
            eng.tune(0, a.remap, 0)
        def split_pairs(n0, pairs):
            for q in range(pairs):
                eng.pair_part(n0 + 2 * q, E.PART_EDGE)
                eng.pair_part(n0 + 2 * q, E.PART_INTERIOR)
        el = timed(eng, split_pairs if a.split else (lambda n0, pairs: eng.step_pairs(n0, pairs)))
    else:
        el = timed(eng, lambda n0, pairs: eng.step(n0, 2 * pairs))
    out[kind] = {"ms_per_step": el * 1e3, "Gcell_per_s": cells / el / 1e9, "gap_GiB": eng.spread_gap_gib()}
    if a.check:
        sums[kind] = [eng.checksum(i) for i in (E.DX, E.DY, E.DZ, E.BX, E.BY, E.BZ)]
    eng.close()
if len(sums) == 2:
    out["checksums_equal"] = sums["single"] == sums["pair"]
print(json.dumps(out))
