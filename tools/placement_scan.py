#!/usr/bin/env python3
"""Step time of the headline workload as a function of where its arena lies inside ONE device allocation:
    FDTD_PLACEMENT_SCAN_MB=0,128,256 python tools/placement_scan.py [edge]
Prints the offsets and the ms/step each one gave (four source-free steps on the zero state)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd_amd  # noqa: E402
from fdtd_amd import engine as E  # noqa: E402
from fdtd_amd import setup as fs  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = fs.constants()
d = (1053e-9 / 20,) * 3
dt = d[0] / (2 * K["c0"])
eng = fdtd_amd.Engine3D((n, n, n), (20,) * 6, d, dt, K["mu0"], 20, 64, precision=32, has_mu=True, mode=E.MODE_FUSED)
offs = os.environ.get("FDTD_PLACEMENT_SCAN_MB", "")
print("offsets MB:", offs)
print("ms/step   :", " ".join(f"{v:.2f}" for v in eng.placement_probes()), flush=True)
