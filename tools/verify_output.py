#!/usr/bin/env python3
"""Checks on a directory of driver output, after the reference's own scripts
(fdtd3d/scripts/verify_vti.py, diagnose_energy.py): every frame finite, last/first energy within
[0.1, 100], and a table of where the energy sits.  Exit status 1 if a check fails.

    python tools/verify_output.py fdtd3d/build/output_3d [--prefix E_mag] [--rim 20]
    python tools/verify_output.py fdtd2d/build/output_2d --csv [--prefix ey]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fdtd_amd import fieldio  # noqa: E402


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("directory")
    ap.add_argument("--prefix", default=None, help="frame family (default E_mag, or ey with --csv)")
    ap.add_argument("--csv", action="store_true", help="2-D CSV frames (needs dims.txt beside them)")
    ap.add_argument("--rim", type=int, default=20, help="cells per face counted as absorber")
    args = ap.parse_args()
    prefix = args.prefix or ("ey" if args.csv else "E_mag")
    files = fieldio.list_frames(args.directory, prefix)
    if not files:
        print(f"no {prefix}_t*.{'csv' if args.csv else 'vti'} files in {args.directory}")
        return 1
    dims = None
    if args.csv:
        for name in ("dims.txt", "dims.txt.gz"):
            if os.path.exists(os.path.join(args.directory, name)):
                dims = fieldio.read_dims2d(os.path.join(args.directory, name))
        if dims is None:
            print("dims.txt missing")
            return 1
    ok, energies = True, []
    print(f"{'step':>7} {'max':>13} {'total':>13} {'centre':>13} {'rim share':>10}")
    for f in files:
        data = fieldio.read_csv2d(f, dims["Nx"], dims["Nz"])[0] if args.csv else fieldio.read_vti(f)[0]
        r = fieldio.energy_report(data, rim=min(args.rim, min(data.shape) // 2))
        if not r["finite"]:
            print(f"{fieldio.frame_step(f):7d} {r['nonfinite']} non-finite values")
            ok = False
            continue
        energies.append(r["total"])
        print(f"{fieldio.frame_step(f):7d} {r['max']:13.6e} {r['total']:13.6e} {r['centre']:13.6e} {r['rim_share']:10.4f}")
    lit = [e for e in energies if e > 0]         # the first frames of a run are dark
    good, ratio = fieldio.check_series(lit)
    print(f"energy ratio last/first lit frame: {ratio:.4g} ({'ok' if good else 'outside [0.1, 100]'})")
    return 0 if (ok and good) else 1


if __name__ == "__main__":
    sys.exit(main())
