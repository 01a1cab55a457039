#!/bin/bash
# Wall time of the C++ fdtd3d driver with the RMS window advanced inside the two-step launches, against one engine
# call per step (what the reference's loop looks like), on the reference's default 200^3 problem and at 1024^3.
#   bash tools/driver_timing.sh > gpurun_out/driver_timing.log
set -e
cd "$(dirname "$0")/.."
exe=fdtd3d/build/fdtd3d
run() { echo "## $*"; ( cd "$(mktemp -d)" && env "$@" FDTD3D_WRITE=0 timeout -k 10 400 "$OLDPWD/$exe" | grep -v "frame dumped" ); }
run FDTD3D_PRECISION=64
run FDTD3D_PRECISION=64 FDTD3D_STEP_BY_STEP=1
run FDTD3D_PRECISION=32
run FDTD3D_PRECISION=32 FDTD3D_STEP_BY_STEP=1
run FDTD3D_NX=1024 FDTD3D_NY=1024 FDTD3D_NZ=1024 FDTD3D_PRECISION=32 FDTD3D_STEPS=101 FDTD3D_FRAME_EVERY=100
run FDTD3D_NX=1024 FDTD3D_NY=1024 FDTD3D_NZ=1024 FDTD3D_PRECISION=32 FDTD3D_STEPS=101 FDTD3D_FRAME_EVERY=100 FDTD3D_STEP_BY_STEP=1
run FDTD3D_NX=1024 FDTD3D_NY=1024 FDTD3D_NZ=1024 FDTD3D_PRECISION=32 FDTD3D_STEPS=101 FDTD3D_FRAME_EVERY=100 FDTD3D_SLABS=2
run FDTD3D_NX=1024 FDTD3D_NY=1024 FDTD3D_NZ=1024 FDTD3D_PRECISION=32 FDTD3D_STEPS=101 FDTD3D_FRAME_EVERY=100 FDTD3D_SLABS=2 FDTD_GROUP_TRANSPORT=rccl
