"""Kernel timeline of the LAST frame of a `rocprofv3 --kernel-trace -- python tools/frame_gaps.py` database: start / end (ms from the frame's
first kernel), grid, name -- what runs beside what when a frame's chunks alternate between two streams.  usage: frame_timeline.py <db> [max rows]"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, start, end, grid_x from kernels order by start").fetchall()
cuts = [i for i in range(len(rows) - 1) if rows[i + 1][1] - max(r[2] for r in rows[: i + 1][-8:]) > 20e6]
last = rows[cuts[-1] + 1:] if cuts else rows
t0 = last[0][1]
span = max(r[2] for r in last) - t0
print("last frame: %d kernels, span %.3f ms" % (len(last), span / 1e6))
for name, s, e, g in last[: int(sys.argv[2]) if len(sys.argv) > 2 else 80]:
    print("%9.3f -> %9.3f  (%8.3f ms)  grid %6d  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, g // 256 if g else 0, name.split("(")[0][:40]))
