#!/bin/bash
for v in sm0 sm1 sm0 sm1; do echo "== $v"; PNR_LIB_PATH=build/ab/libpnr_$v.so timeout 200 python tools/train_profile.py 2>&1 | grep -E "pnr_mlp_wgrad"; done
