"""profiles/rNN_train_summary.txt from the rocprofv3 passes over tools/train_trace.py: per-step kernel table of the kernel trace and
HBM traffic per step from the FETCH_SIZE / WRITE_SIZE passes (FETCH_SIZE doubled per MI355X_MICROARCH.md, WRITE_SIZE as reported).
usage: python tools/train_summary.py <trace.db> <steps in trace> [<fetch.db> <write.db> <steps in the pmc runs>]"""
import sqlite3
import sys

trace, nt = sys.argv[1], int(sys.argv[2])
fetch, write, npmc = (sys.argv[3], sys.argv[4], int(sys.argv[5])) if len(sys.argv) > 5 else (None, None, 1)
cur = sqlite3.connect(trace).cursor()
rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print("== training step (tools/train_trace.py: 4096 rays x (64+192) samples, 8x256 + 45/32 heads, NetworkWrapper losses, Adam), rocprofv3, MI355X")
print(f"total kernel time per step: {tot / nt / 1e6:.3f} ms   (the first step of the trace includes first-launch effects)")
for r in rows[:32]:
    print(f"{r[0][:70]:70s} calls/step {r[1] / nt:5.1f}  ms/step {r[2] / nt / 1e6:7.3f}  avg_us {r[3] / 1e3:8.1f} max_us {r[4] / 1e3:8.1f}")
if fetch is None:
    sys.exit(0)
print()
print("== HBM traffic per step (separate --pmc passes; FETCH_SIZE x2 per MI355X_MICROARCH.md, WRITE_SIZE as reported)")
f = dict(sqlite3.connect(fetch).cursor().execute("select kernel_name, sum(value) from counters_collection where counter_name='FETCH_SIZE' group by kernel_name").fetchall())
w = dict(sqlite3.connect(write).cursor().execute("select kernel_name, sum(value) from counters_collection where counter_name='WRITE_SIZE' group by kernel_name").fetchall())
for k in sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, 0) + w.get(k, 0)))[:10]:
    print(f"{k[:60]:60s} fetch {2 * f.get(k, 0) * 1024 / npmc / 1e9:7.3f} GB  write {w.get(k, 0) * 1024 / npmc / 1e9:7.3f} GB")
