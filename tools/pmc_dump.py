"""Per-kernel sums of every counter in one or more rocprofv3 --pmc databases (rocpd sqlite), normalised per dispatch.
usage: python tools/pmc_dump.py <pmc.db> [<pmc.db> ...] [--like <substring of the kernel name>]..."""
import sqlite3
import sys

dbs = [a for a in sys.argv[1:] if a.endswith(".db")]
likes = [sys.argv[i + 1] for i, a in enumerate(sys.argv) if a == "--like"]
for db in dbs:
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, counter_name, count(*), sum(value), max(value) from counters_collection "
                       "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
    for k, c, n, s, m in rows:
        if likes and not any(l in k for l in likes):
            continue
        print("%-60s %-28s dispatches=%4d  per_dispatch=%.6g  max=%.6g" % (k[:60], c, n, s / n, m))
