#!/usr/bin/env python3
"""Copies what tools/r3_measure.sh left under gpurun_out/ into profiles/ (the newest rocprofv3 output of each directory),
and writes the per-launch tables of the two bench runs.  Runs on the CPU."""
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def newest(pattern):
    files = glob.glob(os.path.join(O, pattern), recursive=True)
    return max(files, key=os.path.getmtime) if files else None


def trace_table(prof_dir, kernel, title, out_name, warm, timed):
    f = newest(f"{prof_dir}/**/*_kernel_trace.csv")
    rows = sorted(((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6) for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"]))
    d = [ms for _, ms in rows]
    per = len(d) // 2                       # the cold-start leg and the noise leg launch the same number of times
    lines = [f"# rocprofv3 --kernel-trace of `{title}`, 1024^3, every launch of the step kernel", "",
             f"The run steps W + K = {warm + timed} steps from the cold start, fills D and B with noise, then W + K = {warm + timed} steps again, of which the last K = {timed} are the timed ones.", "",
             "| launch | state | ms |", "|---|---|---|"]
    n_warm = per * warm // (warm + timed)
    for q, ms in enumerate(d):
        state = "cold start" if q < per else ("noise, warm-up" if q < per + n_warm else "noise, timed")
        lines.append(f"| {q} | {state} | {ms:.3f} |")
    mean = lambda v: sum(v) / max(len(v), 1)
    lines += ["", f"mean of all {len(d)} launches (what `--stats` reports): {mean(d):.3f} ms; cold-start launches: {mean(d[:per]):.3f} ms; "
                  f"timed launches on the noise state: {mean(d[per + n_warm:]):.3f} ms"]
    open(os.path.join(P, out_name), "w").write("\n".join(lines) + "\n")
    shutil.copy(newest(f"{prof_dir}/**/*_kernel_stats.csv"), os.path.join(P, out_name.replace("kernel_trace", "kernel_stats").replace(".md", ".csv")))


trace_table("r3_prof_pair", "k3_pair", "python bench.py` (two steps per launch)", "r3_kernel_trace_bench1024_pair.md", 10, 20)
trace_table("r3_prof_single", "k3_fused", "FDTD_PAIR=0 python bench.py` (one step per launch)", "r3_kernel_trace_bench1024_single.md", 10, 20)
os.replace(os.path.join(P, "r3_kernel_stats_bench1024_single.csv"), os.path.join(P, "r3_kernel_stats_bench1024_single_step.csv"))
for name in ("r3_bench_1024_pair_under_rocprof.json", "r3_bench_1024_single_step_under_rocprof.json"):
    line = [l for l in open(os.path.join(O, name)) if l.startswith("{")][-1]
    open(os.path.join(P, name), "w").write(json.dumps(json.loads(line), indent=1) + "\n")
for name in ("r3_traffic_1024_pair.json", "r3_traffic_1024_fused.json"):
    shutil.copy(os.path.join(O, name), os.path.join(P, name))
shutil.copy(os.path.join(O, "r3_pmc_clock_final.md"), os.path.join(P, "r3_pmc_clock_cold_vs_noise_final.md"))
shutil.copy(os.path.join(O, "r3_group_two_slabs_one_gpu.log"), os.path.join(P, "r3_group_two_slabs_one_gpu.md"))
for name in ("r3_traffic_8192_2d_pair.json", "r3_traffic_8192_2d_fused.json", "r3_bench2d_observed.json"):
    if os.path.exists(os.path.join(O, name)):
        shutil.copy(os.path.join(O, name), os.path.join(P, name))
f2 = newest("r3_prof_2d/**/*_kernel_stats.csv")
if f2:
    shutil.copy(f2, os.path.join(P, "r3_kernel_stats_2d_8192_pair.csv"))
    line = [l for l in open(os.path.join(O, "r3_bench_2d_8192_under_rocprof.json")) if l.startswith("{")][-1]
    open(os.path.join(P, "r3_bench_2d_8192_under_rocprof.json"), "w").write(json.dumps(json.loads(line), indent=1) + "\n")
print("installed")
