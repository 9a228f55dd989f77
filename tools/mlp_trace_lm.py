import json, os, subprocess, sys
ROOT="/root/repo"
def run(lib):
    out = subprocess.run([sys.executable, os.path.join(ROOT,"tools/mlp_trace_fused.py"), "--child", os.path.join(ROOT,"build/ab/libpnr_%s.so"%lib)], capture_output=True, text=True, timeout=240)
    line=[l for l in out.stdout.splitlines() if l.startswith("TRACE ")]
    return json.loads(line[0][6:])
L=run("trL"); M=run("trM")
print("trL build %.3f ms @%d MHz, trM build %.3f ms @%d MHz" % (L["ms"],L["mhz"],M["ms"],M["mhz"]))
tl, tm = L["t"], M["t"]
print(" c   Lw_P  Lw_Q | M_P  M_Q  periodM")
for c in range(1,43):
    print("%2d  %5d %5d | %5d %5d %6d" % (c, tl[0][c][6]-tl[0][c-1][4], tl[4][c][6]-tl[4][c-1][4], tm[0][c][3]-tm[0][c][2], tm[4][c][3]-tm[4][c][2], tm[0][c+1][2]-tm[0][c][2]))
