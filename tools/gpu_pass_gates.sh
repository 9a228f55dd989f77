#!/bin/bash
for v in hd nodma nodmast nost; do echo "== $v"; PNR_MLP_VARIANT=0 PNR_LIB_PATH=build/ab/libpnr_$v.so timeout 200 python tools/clk_probe.py 2>&1 | grep -E "MHz"; done
