#!/usr/bin/env bash
# Configure, build and run fdtd2d the way the reference's driver script does: out-of-tree build in
# ./build, Release, then the executable from inside build/ so that outputs land in build/output_2d.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
mkdir -p "$here/build"
cd "$here/build"
cmake .. -DCMAKE_BUILD_TYPE=Release
make -j"${MAX_JOBS:-8}"
./fdtd2d "$@"
