"""Inside the L phases of the fused ping-pong MLP (-DPNR_TRACE=1 -DPNR_TRACE_MASK=0x53 build): stamps 4 = L start (after the
M->L barrier), 0 = refill_begin, 1 = refill issued (epilogue done), 6 = arrival at the L->M barrier.  Per chunk, wave 0 / wave 4:
  a = 4 -> 0 (what precedes the refill: early fragment requests), b = 0 -> 1 (refill pieces + epilogue [+ side work]),
  c = 1 -> 6 (layer hand-over, late fragment / bias requests, the drain of the phase's LDS reads)
usage: python tools/mlp_trace_l.py <lib name under build/ab>"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(ROOT, "build", "ab", "libpnr_%s.so" % sys.argv[1])
out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mlp_trace_fused.py"), "--child", lib], capture_output=True, text=True, timeout=240)
D = json.loads([l for l in out.stdout.splitlines() if l.startswith("TRACE ")][0][6:])
T = D["t"]
print("trace build: %.3f ms @ %.0f MHz" % (D["ms"], D["mhz"]))
print(" c |  P: a     b     c   total |  Q: a     b     c   total")
for c in range(1, 41):
    row = []
    for w in (0, 4):
        t4, t0, t1, t6 = T[w][c - 1][4], T[w][c - 1][0], T[w][c - 1][1], T[w][c][6]
        row += [t0 - t4, t1 - t0, t6 - t1, t6 - t4]
    print("%2d | %5d %5d %5d %6d | %5d %5d %5d %6d" % tuple([c] + row))
