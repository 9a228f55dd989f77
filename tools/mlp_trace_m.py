"""Inside the M phases of the fused ping-pong MLP (-DPNR_TRACE=1 -DPNR_TRACE_MID=1 -DPNR_TRACE_MASK=0x8f build): stamps 2 = M start,
0 / 7 / 1 = before MFMA NF/4, NF/2, 3NF/4, 3 = all MFMAs issued.  Per chunk, wave 0 / wave 4: the four quarters in cycles.
usage: python tools/mlp_trace_m.py <lib name under build/ab>"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(ROOT, "build", "ab", "libpnr_%s.so" % sys.argv[1])
out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mlp_trace_fused.py"), "--child", lib], capture_output=True, text=True, timeout=240)
D = json.loads([l for l in out.stdout.splitlines() if l.startswith("TRACE ")][0][6:])
T = D["t"]
print("trace build: %.3f ms @ %.0f MHz" % (D["ms"], D["mhz"]))
print(" c |  P: q1    q2    q3    q4  total |  Q: q1    q2    q3    q4  total")
for c in range(0, 41):
    row = []
    for w in (0, 4):
        t2, t0, t7, t1, t3 = (T[w][c][k] for k in (2, 0, 7, 1, 3))
        row += [t0 - t2, t7 - t0, t1 - t7, t3 - t1, t3 - t2]
    print("%2d | %5d %5d %5d %5d %6d | %5d %5d %5d %5d %6d" % tuple([c] + row))
