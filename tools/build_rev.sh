#!/bin/bash
# Build libpnr of a git revision into build/ab/libpnr_<name>.so (same-box A/B against earlier states of the code).
# usage: tools/build_rev.sh <rev> <name> [extra flags]
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d /tmp/pnr_rev.XXXX)
git -C $R archive "$1" panopticnerf_amd/csrc include | tar -x -C $T
mkdir -p $R/build/ab
C=$T/panopticnerf_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -fvisibility=hidden -I$T/include -I$C $3 -shared -o $R/build/ab/libpnr_$2.so $C/*.hip -x hip $C/pnr_api.cpp $C/pnr_mlp_pack.cpp 2>&1 | grep -E "error"
rm -rf $T
ls -la $R/build/ab/libpnr_$2.so
