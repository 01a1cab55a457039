#!/usr/bin/env python3
"""Headline benchmark: fdtd3d cell-update rate on MI355X, weak-scaled z-slabs.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload: 1024 x 1024 x (1024 * N) cells, heterogeneous eps and mu
grids (synthetic, counter hash), 20-layer CPML on all six faces, Gaussian sheet current source,
fp32, one z-slab of 1024 layers per GPU, one ghost plane of tangential D (E) and B (H) exchanged
per step half.  One "step" = E-half + H-half over the whole grid, by default in ONE fused kernel per
step (plus, on a slab below another one, the single-plane H kernel that finishes its top layer).
All inputs are generated on the device before the timed region.  Prints ONE JSON line (rank 0).

Environment knobs for experiments (the defaults are the graded configuration):
  FDTD_BENCH_N=512       edge length per GPU instead of 1024
  FDTD_BENCH_MU=0        constant mu0 instead of a mu grid
  FDTD_BENCH_CPU=0       skip the CPU baseline leg
  FDTD_BENCH_NZ=2048     total layers along z (default: edge length x number of GPUs)
  FDTD_BENCH_BACKEND=gloo, FDTD_BENCH_ONE_DEVICE=1
                         rehearsal of the multi-rank path on a box with one GPU: every rank uses cuda:0
                         and the ghost planes travel through gloo instead of RCCL
  FDTD_BENCH_MODE=lean   two half-step kernels over D,B-only storage instead of the fused default
  FDTD_BENCH_MODE=full   two half-step kernels, E and H resident too (the reference's storage)

Started WITHOUT torch.distributed.run but with --gpus N > 1, the same workload runs as ONE process
driving N GPUs through a slab group (fdtd3d_group_*: streams, events, peer copies instead of RCCL);
the JSON line then says so in config.decomposition.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak


def cpu_baseline():
    """Reference CPU path timed on this box's host cores (rank 0, N=1 only), bounded sample."""
    from oracle import pyoracle as po
    ref_lib = os.path.join(ROOT, "oracle", "_ref", "libref3d.so")
    if os.path.exists(ref_lib):
        # the unmodified reference solver compiled from its own sources: its default 200^3 problem.
        # Its constructor prints a banner on stdout; this program's stdout carries exactly one JSON line,
        # so the banner goes to stderr.
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            r = po.Ref3D()
        finally:
            import ctypes
            ctypes.CDLL(None).fflush(None)       # the banner sits in C stdio's buffer when stdout is a pipe
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
        r.step(0, 2)
        steps = 20
        t0 = time.perf_counter()
        r.step(2, steps)
        el = time.perf_counter() - t0
        cells = 200 ** 3
        return {"value": cells * steps / el / 1e6, "unit": "Mcell-updates/s", "cores": 1, "kind": "reference",
                "sample": f"unmodified reference updateE+updateH, its default 200^3 problem (fp64, 20-layer CPML, "
                          f"eps grid, constant mu0), {steps} steps in {el:.1f} s, single thread as shipped"}
    K = po.consts()
    N, d = (128, 128, 128), (5.265e-8,) * 3
    dt = d[0] / (2 * K["c0"])
    tw = po.Twin3D(64, N, (20,) * 6, d, dt, K["mu0"], 20, 64, True)
    for a in range(3):
        for s in range(2):
            t = po.cpml_face(20, s, d[a], d[a], dt)
            for k in range(6):
                tw.set(po.tab_id(a, s, k), t[k])
    wf, mag = po.default_source3d(N[0], N[1], 64, d, dt)
    tw.set(po.JX_WF, wf)
    tw.set(po.JX_MAG, mag)
    tw.fill_materials(1234, K["eps0"], 4 * K["eps0"], K["mu0"], 2 * K["mu0"])
    tw.step(0, 1)
    steps = 40
    t0 = time.perf_counter()
    tw.step(1, steps)
    el = time.perf_counter() - t0
    return {"value": N[0] * N[1] * N[2] * steps / el / 1e6, "unit": "Mcell-updates/s", "cores": 1, "kind": "port",
            "sample": f"CPU restatement of the reference loops (fp64), 128^3 with eps/mu grids and 20-layer CPML, {steps} steps in {el:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import fdtd_amd
    from fdtd_amd import setup as fs
    from fdtd_amd import slab

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            return main_group(args)
        args.gpus = world
    backend = os.environ.get("FDTD_BENCH_BACKEND", "nccl")
    if os.environ.get("FDTD_BENCH_ONE_DEVICE", "0") == "1":
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    n_edge = int(os.environ.get("FDTD_BENCH_N", "1024"))
    has_mu = os.environ.get("FDTD_BENCH_MU", "1") != "0"
    mode_name = os.environ.get("FDTD_BENCH_MODE", "fused")
    if mode_name not in ("fused", "lean", "full"):
        raise SystemExit(f"FDTD_BENCH_MODE={mode_name}: expected fused, lean or full")
    lean = mode_name == "lean"
    Nx = Ny = n_edge
    Nz = int(os.environ.get("FDTD_BENCH_NZ", str(n_edge * world)))
    npml = (20,) * 6
    K = fs.constants()
    d = (1053e-9 / 20,) * 3                      # the reference's cell: lambda0 / 20
    dt = d[0] / (K["c0"] * 2.0)                  # its CFL choice
    n_total = args.warmup + args.steps
    z_lo, z_hi = slab.partition(Nz, world)[rank]

    eng = fdtd_amd.Engine3D((Nx, Ny, Nz), npml, d, dt, K["mu0"], 20, n_total, precision=32, has_mu=has_mu,
                            mode={"fused": fdtd_amd.engine.MODE_FUSED, "lean": fdtd_amd.engine.MODE_LEAN, "full": fdtd_amd.engine.MODE_FULL}[mode_name],
                            z_range=(z_lo, z_hi), device=local)
    fs.load_cpml3d(eng, d, dt)
    wf, mag = fs.default_source3d(Nx, Ny, n_total, d, dt)
    eng.set(fdtd_amd.engine.JX_WF, wf)
    eng.set(fdtd_amd.engine.JX_MAG, mag)
    eng.fill_materials(1234, K["eps0"], 4.0 * K["eps0"], K["mu0"], 2.0 * K["mu0"])

    stepper = slab.make_gpu_stepper(eng, rank, world) if world > 1 else None
    if stepper is not None:
        stepper.prime()          # RCCL sets its point-to-point channels up here, not in a timed step

    def run(n0, count):
        if stepper is None:
            eng.step(n0, count)
        else:
            stepper.run(n0, count)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(0, args.warmup)
    if stepper is not None:
        stepper.finish()
    fence()
    eng.kernel_timing(True)
    t0 = time.perf_counter()
    run(args.warmup, args.steps)
    if stepper is not None:
        stepper.finish()
    fence()
    el = time.perf_counter() - t0
    eng.kernel_timing(False)
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    # per-kernel device time from HIP events recorded on the launch stream
    n_e, ms_e = eng.kernel_time(0)
    n_h, ms_h = eng.kernel_time(1)
    n_f, ms_f = eng.kernel_time(2)
    cells_local = Nx * Ny * (z_hi - z_lo)
    # algorithmic bytes per cell, fp32, in the reference's formulation whatever the kernels really move:
    # E-half reads H(3) + r/w D(6) + eps(3) + writes E(3) = 15 words; H-half reads E(3) + r/w B(6) + writes
    # H(3) = 12 words, +3 for the mu grid.  A fused launch does both halves of its cells.
    bytes_e, bytes_h = 15 * 4, (12 + (3 if has_mu else 0)) * 4
    steps_ = max(args.steps, 1)
    t_e, t_h, t_f = ms_e / steps_ * 1e-3, ms_h / steps_ * 1e-3, ms_f / steps_ * 1e-3   # device seconds per step per kernel kind
    if mode_name == "fused":
        # per step the fused launches (one, or lowest chunk + rest on a slab) together visit every owned cell once
        dom, dom_bytes, t_dom = "k3_fused", bytes_e + bytes_h, t_f
    elif t_e >= t_h:
        dom, dom_bytes, t_dom = "k3_half_e", bytes_e, t_e
    else:
        dom, dom_bytes, t_dom = "k3_half_h", bytes_h, t_h
    ach = cells_local * dom_bytes / t_dom / 1e9
    sums = [eng.sumsq(i) for i in (fdtd_amd.engine.DX, fdtd_amd.engine.BY)]
    finite = all(v == v and v < float("inf") for v in sums)
    ex_energy = sums[0]
    # order-independent checksum of Dx over the whole grid: equal for every decomposition of the same problem
    csum = eng.checksum(fdtd_amd.engine.DX)
    if world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, csum)
        csum = sum(parts) % 2 ** 64

    # HBM-side bytes per launch from the PMC passes recorded under profiles/ (the counters cannot be
    # read from inside this process); only reported when this run is the configuration they were taken on
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", f"r1_traffic_1024_{mode_name}.json")) as f:
            rec = json.load(f)
        if has_mu and n_edge == 1024 and world == 1:
            traffic = rec["per_launch"][dom]["total_bytes"]
    except (OSError, KeyError, ValueError):
        traffic = None

    if rank == 0:
        total_cells = Nx * Ny * Nz
        value = total_cells * args.steps / el / 1e6
        bytes_step = bytes_e + bytes_h
        out = {
            "metric": "Mcell-updates/s", "value": value, "unit": "Mcell-updates/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"fdtd3d {Nx}x{Ny}x{Nz} ({Nz // world} layers per GPU), heterogeneous eps"
                                   f"{'/mu' if has_mu else ''} grid, 20-layer CPML on six faces, Gaussian sheet source",
                       "grid": [Nx, Ny, Nz], "cpml_layers": 20, "mu_grid": has_mu, "decomposition": f"z-slabs x{world}",
                       "storage": {"fused": "D,B resident in two buffer sets; E=D/eps, H=B/mu formed in registers; one kernel per step",
                                   "lean": "D,B resident; E=D/eps, H=B/mu recomputed in the consuming kernel; two kernels per step",
                                   "full": "D,E,B,H resident; two kernels per step"}[mode_name],
                       "device_bytes_per_gpu": eng.device_bytes(), "placement_probe_ms": [round(v, 3) for v in eng.placement_probes()]},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_rate_GBs": (traffic / t_dom / 1e9) if traffic else None,   # bytes really moved per second
                         "algorithmic_bytes_per_launch": cells_local * dom_bytes,
                         "algorithmic_bytes_per_cell": dom_bytes,
                         "cells_per_launch": cells_local,
                         "ms_per_step_e_half": t_e * 1e3, "ms_per_step_h_half": t_h * 1e3, "ms_per_step_fused": t_f * 1e3,
                         "launches_per_step": {"k3_half_e": n_e / steps_, "k3_half_h": n_h / steps_, "k3_fused": n_f / steps_},
                         "whole_step_frac": (cells_local * world * bytes_step * args.steps / el / 1e9) / (HBM_PEAK_GBS * world)},
            "checks": {"finite": bool(finite), "sumsq_dx_rank0": ex_energy, "checksum_dx": csum},
        }
        if world == 1 and os.environ.get("FDTD_BENCH_CPU", "1") != "0":
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main_group(args):
    """--gpus N in a single process: N slabs, slab r on GPU r, stepped from this one host thread."""
    import fdtd_amd
    from fdtd_amd import engine as E
    from fdtd_amd import setup as fs

    n_edge = int(os.environ.get("FDTD_BENCH_N", "1024"))
    has_mu = os.environ.get("FDTD_BENCH_MU", "1") != "0"
    mode_name = os.environ.get("FDTD_BENCH_MODE", "fused")
    mode = {"fused": E.MODE_FUSED, "lean": E.MODE_LEAN, "full": E.MODE_FULL}[mode_name]
    N = args.gpus
    ndev = E.lib().fdtd_device_count()
    if ndev < N and os.environ.get("FDTD_BENCH_ONE_DEVICE", "0") != "1":
        raise SystemExit(f"--gpus {N} but {ndev} device(s) visible (FDTD_BENCH_ONE_DEVICE=1 puts every slab on device 0)")
    devices = [0] * N if ndev < N else list(range(N))
    Nx = Ny = n_edge
    Nz = int(os.environ.get("FDTD_BENCH_NZ", str(n_edge * N)))
    K = fs.constants()
    d = (1053e-9 / 20,) * 3
    dt = d[0] / (K["c0"] * 2.0)
    n_total = args.warmup + args.steps
    g = E.EngineGroup3D((Nx, Ny, Nz), (20,) * 6, d, dt, K["mu0"], 20, n_total, N, precision=32, has_mu=has_mu, mode=mode, devices=devices)
    wf, mag = fs.default_source3d(Nx, Ny, n_total, d, dt)
    for m in g.members:
        fs.load_cpml3d(m, d, dt)
        m.set(E.JX_WF, wf)
        m.set(E.JX_MAG, mag)
    g.fill_materials(1234, K["eps0"], 4.0 * K["eps0"], K["mu0"], 2.0 * K["mu0"])
    g.step(0, args.warmup)
    g.sync()
    for m in g.members:
        m.kernel_timing(True)
    t0 = time.perf_counter()
    g.step(args.warmup, args.steps)
    g.sync()
    el = time.perf_counter() - t0
    bytes_step = 15 * 4 + (12 + (3 if has_mu else 0)) * 4
    kind = 2 if mode_name == "fused" else 0
    per_member = []
    for m, (lo, hi) in zip(g.members, g.bounds()):
        m.kernel_timing(False)
        _, ms = m.kernel_time(kind)
        if mode_name != "fused":
            ms += m.kernel_time(1)[1]
        per_member.append(Nx * Ny * (hi - lo) * bytes_step / (ms / max(args.steps, 1) * 1e-3) / 1e9 if ms > 0 else 0.0)
    ach = min(per_member)                    # the slowest slab's kernels set the pace
    sums = [g.sumsq(i) for i in (E.DX, E.BY)]
    out = {
        "metric": "Mcell-updates/s", "value": Nx * Ny * Nz * args.steps / el / 1e6, "unit": "Mcell-updates/s", "n_gpus": N,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"fdtd3d {Nx}x{Ny}x{Nz} ({Nz // N} layers per GPU), heterogeneous eps{'/mu' if has_mu else ''} grid, "
                               "20-layer CPML on six faces, Gaussian sheet source",
                   "grid": [Nx, Ny, Nz], "cpml_layers": 20, "mu_grid": has_mu,
                   "decomposition": f"z-slabs x{N}, one process, slab group on devices {devices}", "storage": mode_name},
        "roofline": {"bound": "hbm", "kernel": "k3_fused" if mode_name == "fused" else "k3_half_e+k3_half_h", "achieved": ach,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_cell": bytes_step, "per_slab_GBs": per_member,
                     "whole_step_frac": (Nx * Ny * Nz * bytes_step * args.steps / el / 1e9) / (HBM_PEAK_GBS * len(set(devices)))},
        "checks": {"finite": all(v == v and v < float("inf") for v in sums), "sumsq_dx": sums[0], "checksum_dx": g.checksum(E.DX)},
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
