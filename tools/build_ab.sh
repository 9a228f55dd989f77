#!/bin/bash
# Build A/B variants of libpnr.so into build/ab/ (selected at run time with PNR_LIB_PATH).
# usage: tools/build_ab.sh name1:"-Dflag.." name2:"..."
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/build/ab
C=$R/panopticnerf_amd/csrc
for cfg in "$@"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
    -fvisibility=hidden -I$R/include -I$C $flags -shared -o $R/build/ab/libpnr_$name.so \
    -I$R/build/obj $C/*.hip -x hip $C/pnr_api.cpp $C/pnr_mlp_pack.cpp $C/pnr_mlp_tt.cpp 2>&1 | grep -E "error" &    # build/obj: the two-tile kernel's code object (make all)
done
wait
# the same assembly lint `make all` runs, on the variant's flags (a finding is reported, the library is kept: ablation builds
# break the fetch-marker count on purpose)
for cfg in "$@"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  T=$(mktemp -d)
  for f in pnr_mlp pnr_mlp_bwd pnr_mlp_wgrad; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
      -fvisibility=hidden -I$R/include -I$C $flags -Wno-unused-command-line-argument -S --cuda-device-only -o $T/$f.s $C/$f.hip 2>/dev/null &
  done
  wait
  python3 $R/tools/asm_lint.py $T/*.s && echo "asm_lint $name: clean" || echo "asm_lint $name: FINDINGS (see above)"
  rm -rf $T
done
ls -la $R/build/ab/
