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

roofline.achieved prices the step at the ALGORITHMIC bytes of the reference's formulation (D, E, B, H, eps, mu
all resident: 30 words per cell and step with a mu grid, 27 without), whatever the kernels really move;
roofline.own_model_* prices it at the words the chosen storage mode has to move (fused: 18 / 15), and
roofline.traffic is the HBM traffic a rocprofv3 PMC pass measured on this very build and launch shape (null
when profiles/ holds no pass stamped with this build's kernel sources and launch shape).
For N > 1 the line carries multi_gpu: backend and world size actually seen, and per rank its device, the device
time of every kind of launch, the time its stream sat idle in front of the launches that wait for a ghost plane,
and the host time spent waiting for exchanges.

Started WITHOUT torch.distributed.run but with --gpus N > 1, this program starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process -- before it has touched
a GPU itself -- and relays the child's JSON line, so a plain `python bench.py --gpus N` measures the same
one-process-per-GPU path (two steps per launch on overlapping slabs) as the launcher form.
FDTD_BENCH_GROUP=1 instead runs the workload as ONE process driving N GPUs through a slab group
(fdtd3d_group_*: streams, events, peer copies instead of RCCL); the JSON line then says so in
config.decomposition.

At N = 1 the line also carries `extra`: bounded secondary legs timed by the same program after the headline
engine has been released (FDTD_BENCH_EXTRA=0 skips them): the headline grid in fp64 (the precision that
matches the reference bit for bit), the 512^3 vacuum cavity, and the 8192^2 2-D grid with CPML (reference
slab and dispersive-everywhere materials).
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


KIND_NAME = {0: "half_e", 1: "half_h", 2: "fused", 3: "pair"}
PART_NAME = {0: "all", 1: "edge", 2: "interior", 3: "top"}


def byte_models(has_mu: bool, mode_name: str):
    """(bytes per cell of the E-half, of the H-half) in the reference's formulation, and the bytes per cell
    and step the storage mode has to move at least; fp32."""
    ref_e, ref_h = 15 * 4, (12 + (3 if has_mu else 0)) * 4
    # fused: D, B read and written once, eps (and mu) read once; lean: D (B) read and written by its own half
    # and read by the other, eps and mu read by both halves; full: the reference's traffic
    # pair: the fused kernel's words once per TWO steps
    own_words = {"fused": 15 + (3 if has_mu else 0), "pair": (15 + (3 if has_mu else 0)) / 2, "lean": 18 + (6 if has_mu else 3),
                 "full": 27 + (3 if has_mu else 0)}[mode_name]
    return ref_e, ref_h, own_words * 4


def launch_summary(eng, steps: int):
    """Per step: device milliseconds per kind of launch, and the idle time of the launch stream in front of
    each kind (on a slab the gap in front of an edge launch is the wait for a ghost plane)."""
    tl = eng.kernel_timeline()
    dur, gap, cnt = {}, {}, {}
    prev_end = None
    for kind, part, t0, d in tl:
        name = f"{KIND_NAME.get(kind, kind)}_{PART_NAME.get(part, part)}"
        dur[name] = dur.get(name, 0.0) + d
        cnt[name] = cnt.get(name, 0) + 1
        if prev_end is not None:
            gap[name] = gap.get(name, 0.0) + max(0.0, t0 - prev_end)
        prev_end = t0 + d
    k = max(steps, 1)
    span = (tl[-1][2] + tl[-1][3]) / k if tl else 0.0
    return {"launch_ms_per_step": {n: v / k for n, v in dur.items()}, "launches_per_step": {n: v / k for n, v in cnt.items()},
            "stream_idle_ms_per_step_before": {n: v / k for n, v in gap.items()}, "first_to_last_launch_ms_per_step": span}


def stamped_traffic(eng, mode_name: str, dom: str, applies: bool):
    """HBM bytes per launch of the dominant kernel from the PMC passes committed under profiles/ -- only if
    they were taken on this build (same kernel sources) with this launch shape."""
    if not applies:
        return None, "no PMC pass for this configuration"
    from fdtd_amd.engine import kernel_source_digest
    path = os.path.join(ROOT, "profiles", f"r3_traffic_1024_{mode_name}.json")
    try:
        with open(path) as f:
            rec = json.load(f)
        if rec.get("kernel_source_sha256") != kernel_source_digest():
            return None, "kernel sources changed since the PMC pass"
        if rec.get("launch_shape") != eng.stamp_shape():
            return None, "launch shape differs from the PMC pass"
        return rec["per_launch"][dom]["total_bytes"], os.path.relpath(path, ROOT)
    except (OSError, KeyError, ValueError):
        return None, "no PMC pass recorded"


def live_traffic(dom: str, has_mu: bool):
    """HBM bytes per launch of the dominant kernel measured NOW, on this box and this build: two rocprofv3 --pmc passes
    (FETCH_SIZE, WRITE_SIZE) over a few steps of the same configuration on the non-zero state, run as child processes
    after this process has released its engine (tools/pmc_traffic.py).  None if the profiler is not there or a pass fails."""
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return None, "rocprofv3 not on PATH"
    out = os.path.join(tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp"), "traffic.json")
    steps = "6" if dom == "k3_pair" else "4"
    cmd = [sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), "--out", out, "--kernels", dom, "--",
           "tools/prof3d.py", "--mode", "fused", "--n", "1024", "--steps", steps, "--noise"] + ([] if has_mu else ["--no-mu"])
    try:
        r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=400)
        if r.returncode != 0:
            return None, f"PMC passes failed ({r.returncode}): {(r.stderr or r.stdout)[-200:]}"
        with open(out) as f:
            rec = json.load(f)
        return rec["per_launch"][dom], "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes run by this bench.py as child processes (same build, same box, same configuration on the non-zero state)"
    except Exception as e:          # noqa: BLE001 -- the stamped value stays
        return None, f"{type(e).__name__}: {e}"


def roofline_block(eng, mode_name, has_mu, cells_local, steps, applies_traffic, world, total_cells, el):
    """The roofline object for one engine from its launch timers (call launch_summary first: this consumes them)."""
    n_e, ms_e = eng.kernel_time(0)
    n_h, ms_h = eng.kernel_time(1)
    n_f, ms_f = eng.kernel_time(2)
    n_p, ms_p = eng.kernel_time(3)
    if n_p > 0:
        mode_name = "pair"
    ref_e, ref_h, own = byte_models(has_mu, mode_name)
    k = max(steps, 1)
    t_e, t_h, t_f = ms_e / k * 1e-3, ms_h / k * 1e-3, ms_f / k * 1e-3          # device seconds per step per kernel kind
    t_p = ms_p / k * 1e-3
    if mode_name == "pair":
        # one launch of the two-step kernel is two whole steps of every cell (an odd step count ends with one launch
        # of the one-step kernel: its time is part of the steps' kernel time)
        dom, dom_bytes, t_dom, own_bytes = "k3_pair", ref_e + ref_h, t_p + t_f + t_h, own
    elif mode_name == "fused":
        # per step the fused launches (one; or edge + top + interior on a slab) visit every owned cell once, and on a
        # slab below another one the single-plane H kernel finishes the top layer: all of it is the step's kernel time
        dom, dom_bytes, t_dom, own_bytes = "k3_fused", ref_e + ref_h, t_f + t_h, own
    elif t_e >= t_h:
        dom, dom_bytes, t_dom, own_bytes = "k3_half_e", ref_e, t_e, own // 2
    else:
        dom, dom_bytes, t_dom, own_bytes = "k3_half_h", ref_h, t_h, own // 2
    ach = cells_local * dom_bytes / t_dom / 1e9 if t_dom > 0 else 0.0
    own_rate = cells_local * own_bytes / t_dom / 1e9 if t_dom > 0 else 0.0
    traffic, traffic_src = stamped_traffic(eng, mode_name, dom, applies_traffic)
    spl = 2 if dom == "k3_pair" else 1                    # time steps one launch of the dominant kernel advances
    return {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": traffic_src,
            "traffic_rate_GBs": (traffic / (t_dom * spl) / 1e9) if (traffic and t_dom > 0) else None,   # bytes really moved per second
            "traffic_frac_of_peak": (traffic / (t_dom * spl) / 1e9 / HBM_PEAK_GBS) if (traffic and t_dom > 0) else None,
            "frac_note": "achieved/frac price the step at the reference formulation's bytes (a work rate: it exceeds 1 when the kernels move fewer "
                         "bytes than that formulation); traffic_rate_GBs / traffic_frac_of_peak are the bytes really moved",
            "steps_per_launch": spl,
            "kernel_ms_per_launch": (ms_p / n_p) if n_p > 0 else None,
            "algorithmic_bytes_per_launch": cells_local * dom_bytes * spl, "algorithmic_bytes_per_cell": dom_bytes,
            "own_model_bytes_per_cell": own_bytes, "own_model_GBs": own_rate, "own_model_frac": own_rate / HBM_PEAK_GBS,
            "cells_per_launch": cells_local, "kernel_ms_per_step": t_dom * 1e3,
            "ms_per_step_e_half": t_e * 1e3, "ms_per_step_h_half": t_h * 1e3, "ms_per_step_fused": t_f * 1e3,
            "launches_per_step": {"k3_half_e": n_e / k, "k3_half_h": n_h / k, "k3_fused": n_f / k, "k3_pair": n_p / k},
            "whole_step_frac": (total_cells * (ref_e + ref_h) * steps / el / 1e9) / (HBM_PEAK_GBS * world) if el > 0 else 0.0}


def _timed_steps(torch, eng, warm: int, steps: int):
    """(wall seconds, device seconds of the step kernels, launches per kind of kernel) of `steps` steps after `warm` untimed ones."""
    eng.step(0, warm)
    torch.cuda.synchronize()
    eng.kernel_timing(True)
    t0 = time.perf_counter()
    eng.step(warm, steps)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    eng.kernel_timing(False)
    recs = [eng.kernel_time(kind) for kind in (0, 1, 2, 3)]
    return el, sum(ms for _, ms in recs) * 1e-3, [n for n, _ in recs]


def extra_legs_in_children(scale: str):
    """Every secondary leg in a process of its own, one after the other (this process has released its engine): the
    step time of these kernels depends on where their arrays lie in physical memory, and a process that has held and
    released a hundred GB before does not get the layout a fresh one gets (measured: the 2-D legs 5-10 % slower when
    run inside the process that had timed the headline)."""
    import subprocess
    out = {}
    for name in EXTRA_LEGS[scale]:
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--extra-leg", name, "--extra-scale", scale], cwd=ROOT,
                               capture_output=True, text=True, timeout=280)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                out[name] = {"error": f"leg exited with status {r.returncode}: {r.stderr[-300:]}"}
            else:
                out.update(json.loads(lines[-1]))
        except Exception as e:          # noqa: BLE001 -- reported in the line
            out[name] = {"error": f"{type(e).__name__}: {e}"}
    return out


EXTRA_LEGS = {"full": ["fp64_1024", "cavity_512", "fdtd2d_8192", "fdtd2d_8192_dispersive_everywhere"],
              "small": ["fp64_small", "cavity_small", "fdtd2d_small", "fdtd2d_small_dispersive_everywhere"]}


def extra_legs(torch, scale: str, only=None):
    """Secondary measurements for the one JSON line (N = 1 only, a few seconds each, after the headline engine has
    been released).  `scale` = "full": the sizes the configurations are quoted on; "small": the same code on small
    grids (tests).  `only`: just that leg.  A leg that fails reports its error instead of taking the headline down with it."""
    import numpy as np
    import fdtd_amd
    from fdtd_amd import engine as E
    from fdtd_amd import setup as fs
    K = fs.constants()
    d = (1053e-9 / 20,) * 3
    dt = d[0] / (K["c0"] * 2.0)
    full = scale == "full"
    out = {}

    def leg3d(name, N, npml, precision, has_mu, dipole, warm, steps, words, note):
        if only is not None and name != only:
            return
        try:
            n_total = warm + steps
            eng = fdtd_amd.Engine3D(N, npml, d, dt, K["mu0"], N[2] // 2 if dipole else 20, n_total, precision=precision,
                                    has_mu=has_mu, mode=E.MODE_FUSED, device=torch.cuda.current_device())
            try:
                if any(npml):
                    fs.load_cpml3d(eng, d, dt)
                wf, mag = fs.default_source3d(N[0], N[1], n_total, d, dt)
                if dipole:
                    mag[:] = 0
                    mag[N[0] // 2, N[1] // 2] = 1.0
                eng.set(E.JX_WF, wf)
                eng.set(E.JX_MAG, mag)
                if dipole:
                    eng.fill_materials(1234, K["eps0"], K["eps0"], K["mu0"], K["mu0"])
                else:
                    eng.fill_materials(1234, K["eps0"], 4.0 * K["eps0"], K["mu0"], 2.0 * K["mu0"])
                for c in range(3):        # timed on a state without zeros, like the headline
                    eng.fill_array(E.DX + c, 7, -1e-11, 1e-11)
                    eng.fill_array(E.BX + c, 7, -1e-8, 1e-8)
                el, dev_s, _ = _timed_steps(torch, eng, warm, steps)
                cells = N[0] * N[1] * N[2]
                bpc = words * (precision // 8)              # the reference formulation's bytes per cell and step
                ss, cs = eng.sumsq(E.DX), eng.checksum(E.DX)
                out[name] = {"workload": note, "state": "every cell of D and B non-zero (counter-hash noise)", "grid": list(N), "dtype": f"f{precision}", "steps": steps, "warmup": warm,
                             "ms_per_step": el / steps * 1e3, "value": cells * steps / el / 1e6, "unit": "Mcell-updates/s",
                             "kernel_ms_per_step": dev_s / steps * 1e3, "steps_per_launch": 2 if eng.pair_info()["used_by_step"] else 1,
                             "roofline": {"bound": "hbm", "algorithmic_bytes_per_cell": bpc, "achieved": cells * bpc * steps / dev_s / 1e9,
                                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": cells * bpc * steps / dev_s / 1e9 / HBM_PEAK_GBS},
                             "device_bytes": eng.device_bytes(), "physical_gap_between_arrays_gib": eng.spread_gap_gib(),
                             "checks": {"finite": bool(np.isfinite(ss)), "sumsq_dx": ss, "checksum_dx": cs}}
            finally:
                eng.close()
        except Exception as e:          # noqa: BLE001 -- reported in the line
            out[name] = {"error": f"{type(e).__name__}: {e}"}

    def leg2d(name, N, material, warm, steps):
        if only is not None and name != only:
            return
        try:
            dx = K["c0"] / 1.5e14 / 40
            dt2 = dx / (2 * K["c0"])
            eng = fdtd_amd.Engine2D(N, N, (20,) * 4, dx, dx, dt2, K["mu0"], K["eps0"], max(N // 2 - 40, 1), N // 2, steps + warm,
                                    precision=32, mode=E.MODE2D_FUSED, device=torch.cuda.current_device())
            try:
                fs.load_cpml2d(eng, dx, dx, dt2)
                eng.set(E.D2_JY, fs.default_source2d(steps + warm, dx, dx, dt2))
                coef = fs.default_lorentz2d(N, N, dt2)
                if material == "everywhere":
                    coef = [np.full((N, N), c[N // 2, 0]) for c in coef]
                n_disp = int(np.count_nonzero((coef[0] != 0) | (coef[1] != 0) | (coef[2] != 0)))
                for q, c in enumerate(coef):
                    eng.set(E.D2_C1 + q, c)
                del coef
                el, dev_s, launches = _timed_steps(torch, eng, warm, steps)
                cells = N * N
                alg = n_disp * 132 + (cells - n_disp) * 24      # dispersive cell 33 words, plain cell 6 words (fp32)
                ss = eng.sumsq(E.D2_EY)
                out[name] = {"workload": f"fdtd2d {N}x{N}, 20-layer CPML on four sides, "
                                         + ("reference Lorentz slab (41 columns), plain elsewhere" if material == "slab" else "Lorentz-dispersive at every cell"),
                             "grid": [N, N], "dtype": "f32", "steps": steps, "warmup": warm, "ms_per_step": el / steps * 1e3,
                             "value": cells * steps / el / 1e6, "unit": "Mcell-updates/s", "kernel_ms_per_step": dev_s / steps * 1e3,
                             "dispersive_cells": n_disp, "steps_per_launch": 2 if launches[3] else 1,
                             "roofline": {"bound": "hbm", "kernel": "k2_pair" if launches[3] else "k2_fused", "algorithmic_bytes_per_step": alg,
                                          "algorithmic_bytes_per_dispersive_cell": 132, "algorithmic_bytes_per_plain_cell": 24,
                                          "achieved": alg * steps / dev_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": alg * steps / dev_s / 1e9 / HBM_PEAK_GBS},
                             "checks": {"finite": bool(np.isfinite(ss)), "sumsq_ey": ss}}
            finally:
                eng.close()
        except Exception as e:          # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {e}"}

    n3, nc, n2 = (1024, 512, 8192) if full else (96, 64, 512)
    leg3d("fp64_1024" if full else "fp64_small", (n3,) * 3, (20,) * 6 if full else (6,) * 6, 64, True, False, 4, 10, 30,
          f"the headline configuration in fp64, the precision in which the engine equals the reference bit for bit: {n3}^3, eps/mu grids, CPML on six faces")
    leg3d("cavity_512" if full else "cavity_small", (nc,) * 3, (0,) * 6, 32, False, True, 10, 100 if full else 10, 27,
          f"vacuum cavity {nc}^3, PEC walls (no CPML), one-cell Jx dipole at the centre, constant mu0")
    leg2d("fdtd2d_8192" if full else "fdtd2d_small", n2, "slab", 10, 100 if full else 10)
    leg2d("fdtd2d_8192_dispersive_everywhere" if full else "fdtd2d_small_dispersive_everywhere", n2, "everywhere", 10, 60 if full else 10)
    return out


def relaunch_one_process_per_gpu(args) -> int:
    """`python bench.py --gpus N` without a launcher: run the launcher form as a child and relay its one JSON
    line.  Nothing in this process has touched a GPU yet (torch is not even imported), so starting a child is safe."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    if env.get("FDTD_BENCH_ONE_DEVICE", "0") == "1":
        env.setdefault("FDTD_BENCH_BACKEND", "gloo")      # RCCL refuses two ranks on one device
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup)]
    child = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, text=True)
    lines = []
    for ln in child.stdout:
        if ln.startswith("{"):
            lines.append(ln)
        else:
            sys.stderr.write(ln)          # launcher chatter stays off this program's stdout: one JSON line only
    rc = child.wait()
    for ln in lines[-1:]:
        rec = json.loads(ln)
        rec.setdefault("config", {})["launched_as"] = "child torch.distributed.run of a plain `bench.py --gpus N`"
        print(json.dumps(rec))
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--extra-leg", default=None, help=argparse.SUPPRESS)       # one secondary leg in this process (see extra_legs_in_children)
    ap.add_argument("--extra-scale", default="full", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.extra_leg:
        import torch
        torch.cuda.set_device(0)
        print(json.dumps(extra_legs(torch, args.extra_scale, only=args.extra_leg)))
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and os.environ.get("FDTD_BENCH_GROUP", "0") != "1":
        sys.exit(relaunch_one_process_per_gpu(args))

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
    Nx = Ny = n_edge
    Nz = int(os.environ.get("FDTD_BENCH_NZ", str(n_edge * world)))
    npml = (20,) * 6
    K = fs.constants()
    d = (1053e-9 / 20,) * 3                      # the reference's cell: lambda0 / 20
    dt = d[0] / (K["c0"] * 2.0)                  # its CFL choice
    n_total = args.warmup + args.steps
    parts = slab.partition(Nz, world)
    z_lo, z_hi = parts[rank]
    # several ranks, fused storage: two steps per launch on slabs that overlap by one layer either side of a cut
    # (slab.PairSlabStepper) unless FDTD_BENCH_STEPPER=single asks for the one-step kernels with their per-step exchanges
    pair_slabs = (world > 1 and mode_name == "fused" and os.environ.get("FDTD_BENCH_STEPPER", "pair") != "single"
                  and os.environ.get("FDTD_PAIR", "1") != "0" and slab.pair_slabs_possible(parts, Nz, (npml[4], npml[5])))
    stored = slab.pair_slab_range(z_lo, z_hi, Nz) if pair_slabs else (z_lo, z_hi)

    eng = fdtd_amd.Engine3D((Nx, Ny, Nz), npml, d, dt, K["mu0"], 20, 2 * n_total, precision=32, has_mu=has_mu,
                            mode={"fused": fdtd_amd.engine.MODE_FUSED, "lean": fdtd_amd.engine.MODE_LEAN, "full": fdtd_amd.engine.MODE_FULL}[mode_name],
                            z_range=stored, device=local)
    fs.load_cpml3d(eng, d, dt)
    wf, mag = fs.default_source3d(Nx, Ny, 2 * n_total, d, dt)
    eng.set(fdtd_amd.engine.JX_WF, wf)
    eng.set(fdtd_amd.engine.JX_MAG, mag)
    eng.fill_materials(1234, K["eps0"], 4.0 * K["eps0"], K["mu0"], 2.0 * K["mu0"])

    if world == 1:
        stepper = None
    elif pair_slabs:
        stepper = slab.make_gpu_pair_stepper(eng, rank, world, (z_lo, z_hi))
    else:
        stepper = slab.make_gpu_stepper(eng, rank, world)
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

    # The step time of the two-step kernel depends on the data (measured at 1024^3: 11.09 ms from the cold start, where
    # nearly every cell still holds zero; 11.46 with the wave across half the grid; 11.57 with noise in every cell), so
    # the timed state is the one a run spends its life in: every cell of D and B non-zero.  The cold start is timed
    # first, the same W + K steps, and reported beside it (`cold_start`).
    cold = None
    if os.environ.get("FDTD_BENCH_STATE", "noise") != "cold":
        run(0, args.warmup)
        if stepper is not None:
            stepper.finish()
        fence()
        t0 = time.perf_counter()
        run(args.warmup, args.steps)
        if stepper is not None:
            stepper.finish()
        fence()
        cold = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([cold], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            cold = float(t.item())
        # counter hash of the canonical cell index: the same values whatever the decomposition, ghost planes included
        for c in range(3):
            eng.fill_array(fdtd_amd.engine.DX + c, 7, -1e-11, 1e-11)
            eng.fill_array(fdtd_amd.engine.BX + c, 7, -1e-8, 1e-8)
        n_first = n_total
    else:
        n_first = 0
    run(n_first, args.warmup)
    if stepper is not None:
        stepper.finish()
    fence()
    eng.kernel_timing(True)
    slab.WAIT_CLOCK.update(seconds=0.0, calls=0)
    t0 = time.perf_counter()
    run(n_first + args.warmup, args.steps)
    if stepper is not None:
        stepper.finish()
    t_issue = time.perf_counter() - t0          # host time until every launch and exchange of the timed steps was issued
    fence()
    el_own = time.perf_counter() - t0
    eng.kernel_timing(False)
    el = el_own
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    # per-launch device time from HIP events recorded on the launch stream
    cells_local = Nx * Ny * (z_hi - z_lo)
    launches = launch_summary(eng, args.steps)
    roof = roofline_block(eng, mode_name, has_mu, cells_local, args.steps, has_mu and n_edge == 1024 and world == 1,
                          world, Nx * Ny * Nz, el)
    # reductions over the layers this rank answers for (an overlapping slab stores one more either side)
    top = Nz + 1 if rank == world - 1 else z_hi
    sums, csum = [], 0
    for i in (fdtd_amd.engine.DX, fdtd_amd.engine.BY):
        ss, cs = eng.reduce_range(i, z_lo, top)
        sums.append(ss)
        if i == fdtd_amd.engine.DX:
            csum = cs                     # order-independent checksum of Dx: equal for every decomposition of the same problem
    finite = all(v == v and v < float("inf") for v in sums)
    ex_energy = sums[0]
    multi = None
    if world > 1:
        k = max(args.steps, 1)
        mine = {"rank": rank, "device": torch.cuda.current_device(), "device_name": torch.cuda.get_device_name(),
                "layers": [z_lo, z_hi], "stored_layers": list(stored), "stepper": type(stepper).__name__, "wall_ms_per_step": el_own / k * 1e3, "host_issue_ms_per_step": t_issue / k * 1e3,
                "host_wait_ms_per_step": slab.WAIT_CLOCK["seconds"] / k * 1e3, "exchange_waits_per_step": slab.WAIT_CLOCK["calls"] / k,
                "kernel_ms_per_step": roof["kernel_ms_per_step"], "physical_gap_per_array_gib": [round(v, 2) for v in eng.spread_gaps()],
                "checksum_dx": csum, **launches}
        parts = [None] * world
        dist.all_gather_object(parts, mine)
        csum = sum(p["checksum_dx"] for p in parts) % 2 ** 64
        multi = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "per_rank": parts,
                 "slowest_rank": max(parts, key=lambda p: p["wall_ms_per_step"])["rank"]}

    if rank == 0:
        total_cells = Nx * Ny * Nz
        value = total_cells * args.steps / el / 1e6
        out = {
            "metric": "Mcell-updates/s", "value": value, "unit": "Mcell-updates/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"fdtd3d {Nx}x{Ny}x{Nz} ({Nz // world} layers per GPU), heterogeneous eps"
                                   f"{'/mu' if has_mu else ''} grid, 20-layer CPML on six faces, Gaussian sheet source",
                       "grid": [Nx, Ny, Nz], "cpml_layers": 20, "mu_grid": has_mu, "decomposition": f"z-slabs x{world}",
                       "storage": {"fused": "D,B resident in two buffer sets; E=D/eps, H=B/mu formed in registers; "
                                            + ("one kernel per TWO steps (k3_pair)" if roof["kernel"] == "k3_pair" else "one kernel per step"),
                                   "lean": "D,B resident; E=D/eps, H=B/mu recomputed in the consuming kernel; two kernels per step",
                                   "full": "D,E,B,H resident; two kernels per step"}[mode_name],
                       "parity_mode": "fp32 (as the metric is worded): equals an fp32 restatement of the reference's loops bit for bit; "
                                      "within 1e-6 relative L2 of the fp64 reference for about 700 steps of its default 3-D problem "
                                      "(4e-8*sqrt(steps)).  precision=64 equals the reference bit for bit over its whole run: extra.fp64_1024",
                       "launch_shape": eng.stamp_shape(),
                       "device_bytes_per_gpu": eng.device_bytes(), "physical_gap_per_array_gib": [round(v, 2) for v in eng.spread_gaps()],
                       "physical_gap_between_arrays_gib": eng.spread_gap_gib()},
            "roofline": roof,
            "launches": launches,
            "state": "every cell of D and B non-zero (counter-hash noise, |D| < 1e-11, |B| < 1e-8) before the W warm-up steps" if cold is not None
                     else "cold start: fields zero before the W warm-up steps",
            "cold_start": None if cold is None else {"ms_per_step": cold / args.steps * 1e3, "value": total_cells * args.steps / cold / 1e6,
                                                     "state": "fields zero before W warm-up steps; after W + K steps the wave has reached a few planes"},
            "checks": {"finite": bool(finite), "sumsq_dx_rank0": ex_energy, "checksum_dx": csum},
        }
        if multi is not None:
            out["multi_gpu"] = multi
        extra_scale = os.environ.get("FDTD_BENCH_EXTRA", "full" if (n_edge == 1024 and mode_name == "fused") else "0")
        want_pmc = world == 1 and n_edge == 1024 and mode_name == "fused" and os.environ.get("FDTD_BENCH_PMC", "1") != "0"
        if world == 1 and (extra_scale in ("full", "small", "1") or want_pmc):
            eng.close()                   # the children need the memory (fp64 at 1024^3 holds 161 GB)
        if want_pmc:
            rec, src = live_traffic(roof["kernel"], has_mu)
            if rec is not None:
                t_launch = roof["kernel_ms_per_step"] * 1e-3 * roof["steps_per_launch"]
                roof.update({"traffic": rec["total_bytes"], "traffic_read_bytes": rec["read_bytes"], "traffic_write_bytes": rec["write_bytes"],
                             "traffic_source": src, "traffic_rate_GBs": rec["total_bytes"] / t_launch / 1e9,
                             "traffic_frac_of_peak": rec["total_bytes"] / t_launch / 1e9 / HBM_PEAK_GBS})
            else:
                roof["traffic_live_error"] = src
        if world == 1 and extra_scale in ("full", "small", "1"):
            out["extra"] = extra_legs_in_children("small" if extra_scale == "small" else "full")
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
    g = E.EngineGroup3D((Nx, Ny, Nz), (20,) * 6, d, dt, K["mu0"], 20, 2 * n_total, N, precision=32, has_mu=has_mu, mode=mode, devices=devices)
    wf, mag = fs.default_source3d(Nx, Ny, 2 * n_total, d, dt)
    for m in g.members:
        fs.load_cpml3d(m, d, dt)
        m.set(E.JX_WF, wf)
        m.set(E.JX_MAG, mag)
    g.fill_materials(1234, K["eps0"], 4.0 * K["eps0"], K["mu0"], 2.0 * K["mu0"])
    # (the same sequence as the one-process-per-GPU form: W + K steps from the cold start, then the timed state with
    # every cell of D and B non-zero)
    n_first = 0
    if os.environ.get("FDTD_BENCH_STATE", "noise") != "cold":
        g.step(0, n_total)
        g.sync()
        for m in g.members:
            for c in range(3):
                m.fill_array(E.DX + c, 7, -1e-11, 1e-11)
                m.fill_array(E.BX + c, 7, -1e-8, 1e-8)
        n_first = n_total
    g.step(n_first, args.warmup)
    g.sync()
    for m in g.members:
        m.kernel_timing(True)
    t0 = time.perf_counter()
    g.step(n_first + args.warmup, args.steps)
    t_issue = time.perf_counter() - t0
    g.sync()
    el = time.perf_counter() - t0
    per_slab, roofs = [], []
    for r, (m, (lo, hi)) in enumerate(zip(g.members, g.bounds())):
        m.kernel_timing(False)
        launches = launch_summary(m, args.steps)
        roof = roofline_block(m, mode_name, has_mu, Nx * Ny * (hi - lo), args.steps, False, len(set(devices)), Nx * Ny * Nz, el)
        roofs.append(roof)
        per_slab.append({"slab": r, "device": devices[r], "layers": [lo, hi], "kernel_ms_per_step": roof["kernel_ms_per_step"],
                         "physical_gap_per_array_gib": [round(v, 2) for v in m.spread_gaps()], **launches})
    roof = min(roofs, key=lambda q: q["achieved"])          # the slowest slab's kernels set the pace
    roof["per_slab_GBs"] = [q["achieved"] for q in roofs]
    sums = [g.sumsq(i) for i in (E.DX, E.BY)]
    out = {
        "metric": "Mcell-updates/s", "value": Nx * Ny * Nz * args.steps / el / 1e6, "unit": "Mcell-updates/s", "n_gpus": N,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"fdtd3d {Nx}x{Ny}x{Nz} ({Nz // N} layers per GPU), heterogeneous eps{'/mu' if has_mu else ''} grid, "
                               "20-layer CPML on six faces, Gaussian sheet source",
                   "grid": [Nx, Ny, Nz], "cpml_layers": 20, "mu_grid": has_mu, "launch_shape": g.members[0].launch_shape(),
                   "decomposition": f"z-slabs x{N}, one process, slab group on devices {devices}", "storage": mode_name},
        "roofline": roof,
        "multi_gpu": {"backend": "one host thread: HIP streams and events, hipMemcpyPeerAsync between neighbouring devices",
                      "world_size": N, "devices_visible": ndev, "host_issue_ms_per_step": t_issue / max(args.steps, 1) * 1e3,
                      "per_rank": per_slab},
        "checks": {"finite": all(v == v and v < float("inf") for v in sums), "sumsq_dx": sums[0], "checksum_dx": g.checksum(E.DX)},
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
