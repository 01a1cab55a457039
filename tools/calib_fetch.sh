#!/bin/bash
# FETCH_SIZE (and the request counters behind it) of tools/calib_fetch's kernels: one rocprofv3 pass per counter set,
# then a table of counter value x 1024 (FETCH_SIZE is in KiB) next to the byte counts the program prints.
#     hipcc -O3 --offload-arch=gfx950 tools/calib_fetch.hip -o tools/calib_fetch      (beforehand, on the CPU)
#     bash tools/calib_fetch.sh gpurun_out/calib
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/${1:-gpurun_out/calib}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 60 "$ROOT/tools/calib_fetch" > "$OUT/plain.txt"
cat "$OUT/plain.txt"
timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- "$ROOT/tools/calib_fetch" > "$OUT/fetch.txt" 2>&1
timeout -s KILL 90 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d "$OUT/rdreq" -- "$ROOT/tools/calib_fetch" > "$OUT/rdreq.txt" 2>&1 || echo "request-counter pass failed (see rdreq.txt)"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
acc = {}
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        v = acc.setdefault((k, row["Counter_Name"]), [0.0, 0])
        v[0] += float(row["Counter_Value"]); v[1] += 1
with open(os.path.join(out, "counters.txt"), "w") as fh:
    for (k, c), (v, n) in sorted(acc.items()):
        line = f"{k:28s} {c:24s} per launch {v / n:16.1f}" + (f"  = {v / n * 1024 / 1e6:9.1f} MB as KiB" if c == "FETCH_SIZE" else "")
        print(line); fh.write(line + "\n")
PY
