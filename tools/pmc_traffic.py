#!/usr/bin/env python3
"""HBM-side bytes per kernel launch from rocprofv3 PMC counters, stamped with the build they were taken on.

    python tools/pmc_traffic.py --out profiles/r2_traffic_1024_fused.json --kernels k3_fused -- tools/prof3d.py --mode fused --n 1024 --steps 3

Runs the program after `--` twice under `rocprofv3 --pmc`, once for FETCH_SIZE and once for WRITE_SIZE (the
two do not fit one pass), with nothing else traced.  Units and corrections: both counters are in KiB;
on gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes, so read bytes =
2 x FETCH_SIZE x 1024; WRITE_SIZE is exact for 16-byte-per-lane stores.  Infinity-Cache hits are counted
as traffic.  The JSON records the sha256 of the kernel sources and the launch shape the profiled program
reports (it writes them to the file named by FDTD_SHAPE_OUT), so that bench.py only quotes the numbers
next to timings of the same build.  This driver never touches the GPU itself."""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one_pass(counter, prog, workdir, env):
    out = os.path.join(workdir, counter)
    cmd = ["timeout", "-s", "KILL", "150", "rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", out, "--", "python3"] + prog
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit(f"{counter} pass failed ({r.returncode}):\n{r.stdout[-1500:]}\n{r.stderr[-1500:]}")
    acc = {}
    for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                v = acc.setdefault(row["Kernel_Name"], [0.0, 0])
                v[0] += float(row["Counter_Value"])
                v[1] += 1
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--kernels", default="k3_,k2_", help="comma-separated substrings of the kernel names to keep")
    ap.add_argument("--workload", default="")
    ap.add_argument("--variants", action="store_true", help="keep the template arguments in the kernel names (one record per instantiation)")
    ap.add_argument("prog", nargs=argparse.REMAINDER)
    args = ap.parse_args()
    prog = [p for p in args.prog if p != "--"]
    prog[0] = os.path.join(ROOT, prog[0]) if not os.path.isabs(prog[0]) else prog[0]
    from fdtd_amd.engine import kernel_source_digest
    work = tempfile.mkdtemp(prefix="pmc_", dir="/tmp")
    shape_file = os.path.join(work, "shape.json")
    env = dict(os.environ, TMPDIR="/tmp", FDTD_SHAPE_OUT=shape_file)
    fetch = one_pass("FETCH_SIZE", prog, work, env)
    write = one_pass("WRITE_SIZE", prog, work, env)
    keep = [k for k in args.kernels.split(",") if k]
    per = {}
    for name in sorted(set(fetch) | set(write)):
        if not any(k in name for k in keep):
            continue
        short = (name.split("(")[0] if args.variants else name.split("<")[0].split("(")[0]).replace("void ", "").replace("fdtd::", "").strip()
        f, nf = fetch.get(name, [0.0, 0])
        w, nw = write.get(name, [0.0, 0])
        rec = per.setdefault(short, {"launches": 0, "read_bytes": 0.0, "write_bytes": 0.0})
        rec["launches"] += max(nf, nw)
        rec["read_bytes"] += 2.0 * f * 1024.0
        rec["write_bytes"] += w * 1024.0
    for rec in per.values():
        n = max(rec["launches"], 1)
        rec["read_bytes"] /= n
        rec["write_bytes"] /= n
        rec["total_bytes"] = rec["read_bytes"] + rec["write_bytes"]
    shape = None
    if os.path.exists(shape_file):
        with open(shape_file) as fh:
            shape = json.load(fh)
    doc = {"workload": args.workload or " ".join(os.path.relpath(p, ROOT) if os.path.isabs(p) else p for p in prog),
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; read bytes = 2 x FETCH_SIZE x 1024 (gfx950), "
                     "write bytes = WRITE_SIZE x 1024; per-launch means",
           "kernel_source_sha256": kernel_source_digest(), "launch_shape": shape, "per_launch": per}
    with open(os.path.join(ROOT, args.out) if not os.path.isabs(args.out) else args.out, "w") as fh:
        json.dump(doc, fh, indent=1)
    print(json.dumps(doc["per_launch"], indent=1))


if __name__ == "__main__":
    main()
