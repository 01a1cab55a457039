#!/usr/bin/env python3
"""Shader clock and instruction counts of the step kernels under rocprofv3 PMC, for the cold-start state and for a
state with noise in every cell: cycles the GPU was active (GRBM_GUI_ACTIVE) over the kernel's duration give the clock
it really ran at; the SQ counters say whether the same instructions were issued.

    python tools/pmc_clock.py [--n 1024] [--steps 6] > gpurun_out/pmc_clock.md
This driver never touches the GPU itself."""
import argparse
import csv
import glob
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1024)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--counters", default="GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU")
args = ap.parse_args()
work = tempfile.mkdtemp(prefix="pmcclk_", dir="/tmp")
env = dict(os.environ, TMPDIR="/tmp")
print("| state | kernel | launches | ms/launch | " + " | ".join(args.counters.split()) + " | MHz (GRBM_GUI_ACTIVE / duration) |")
print("|---|---|---|---|" + "---|" * (len(args.counters.split()) + 1))
for state in ("cold", "noise"):
    out = os.path.join(work, state)
    prog = [os.path.join(ROOT, "tools", "prof3d.py"), "--mode", "fused", "--n", str(args.n), "--steps", str(args.steps)] + (["--noise"] if state == "noise" else [])
    cmd = ["timeout", "-s", "KILL", "200", "rocprofv3", "--kernel-trace", "--pmc"] + args.counters.split() + ["--output-format", "csv", "-d", out, "--", "python3"] + prog
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit(f"{state} pass failed ({r.returncode}):\n{r.stdout[-1500:]}\n{r.stderr[-1500:]}")
    acc = {}
    for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"]
                if "k3_" not in name:
                    continue
                short = name.split("<")[0].split("(")[0].replace("void ", "").replace("fdtd::", "").strip()
                rec = acc.setdefault(short, {"n": set(), "dur": {}, "ctr": {}})
                did = row.get("Dispatch_Id", row.get("Correlation_Id"))
                rec["n"].add(did)
                if "Start_Timestamp" in row and row["Start_Timestamp"]:
                    rec["dur"][did] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6
                rec["ctr"][row["Counter_Name"]] = rec["ctr"].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    durs = {}
    for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"]
                if "k3_" not in name:
                    continue
                short = name.split("<")[0].split("(")[0].replace("void ", "").replace("fdtd::", "").strip()
                durs.setdefault(short, []).append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6)
    for short, rec in sorted(acc.items()):
        n = max(len(rec["n"]), 1)
        d = durs.get(short) or list(rec["dur"].values())
        ms = sum(d) / len(d) if d else float("nan")
        cells = [f"{rec['ctr'].get(c, float('nan')) / n:.4g}" for c in args.counters.split()]
        mhz = rec["ctr"].get("GRBM_GUI_ACTIVE", float("nan")) / n / (ms * 1e-3) / 1e6 if d else float("nan")
        print(f"| {state} | {short} | {n} | {ms:.3f} | " + " | ".join(cells) + f" | {mhz:.0f} |", flush=True)
