#!/bin/bash
# A/B build of libpnr.so with a variant of the two-tile assembly kernel: tools/build_tt_variant.sh <name> [VAR=value ...]
# (generator environment, e.g. PNR_TT_DMA_POLICY="" for the default cache policy of the weight pieces) -> build/ab/libpnr_<name>.so
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/panopticnerf_amd/csrc; name=$1; shift
D=$R/build/ab_tt/$name; mkdir -p $D $R/build/ab
L=/opt/rocm/lib/llvm/bin
env "$@" python3 $C/asm/gen_mlp_tt.py $D/pnr_mlp_tt.s $D/units.txt || exit 1
$L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $D/pnr_mlp_tt.s -o $D/tt.o && $L/ld.lld -shared $D/tt.o -o $D/pnr_mlp_tt.co || exit 1
python3 -c "import sys; d = open(sys.argv[1], 'rb').read(); open(sys.argv[2], 'w').write(','.join(str(b) for b in d) + '\n')" $D/pnr_mlp_tt.co $D/pnr_mlp_tt_co.inc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fvisibility=hidden \
  -I$R/include -I$C -I$D -shared -o $R/build/ab/libpnr_$name.so $C/*.hip -x hip $C/pnr_api.cpp $C/pnr_mlp_pack.cpp $C/pnr_mlp_tt.cpp 2>&1 | grep -E "error"
ls -la $R/build/ab/libpnr_$name.so
