"""What the part's power limit makes of operand toggling: the register-only bf16 MFMA loop on random operands, with the order in which
the A (weight fragment) and B (activation slice) operands change from MFMA to MFMA as the variable (pnrb_probe_mfma_order).
   python tools/mfma_order_probe.py [repeats]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from panopticnerf_amd import benchlib

NAMES = {0: "A and B change every MFMA", 1: "(b0,t0) (b0,t1) (b1,t0) (b1,t1)  two-tile kernel's order", 2: "(b0,t0) (b0,t1) (b1,t1) (b1,t0)  snake",
         3: "only A changes", 4: "only B changes"}
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for rep in range(reps):
    for p in (0, 1, 2, 3, 4):
        tf, mhz = benchlib.probe_mfma_order(p, 12000)
        print("rep %d  pattern %d  %-58s %7.1f TFLOP/s at %6.0f MHz" % (rep, p, NAMES[p], tf, mhz), flush=True)
pk_c, mhz_c = benchlib.probe_mfma_peak(False, 12000)
print("constant operands (k_mfma_peak<false>)                                      %7.1f TFLOP/s at %6.0f MHz" % (pk_c, mhz_c))
