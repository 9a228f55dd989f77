"""Per-kernel table of rocprofv3 --pmc passes (rocpd sqlite files under a directory): the largest dispatch's value of every
counter found, for kernels matching the given substrings.  usage: python tools/pmc_table.py <dir> [substr ...]"""
import glob, os, sqlite3, sys
root = sys.argv[1]
pats = sys.argv[2:] or ["k_mlp_bwd", "k_mlp_fused", "k_mlp_pp", "k_wgrad("]
tab = {}
for db in glob.glob(os.path.join(root, "**", "*.db"), recursive=True):
    cur = sqlite3.connect(db).cursor()
    try:
        rows = cur.execute("select kernel_name, counter_name, max(value), avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
    except sqlite3.Error as e:
        print("skip", db, e); continue
    for k, c, mx, av, n in rows:
        for p in pats:
            if p in k:
                tab.setdefault(p, {})[c] = (mx, av, n)
for p in pats:
    if p not in tab: continue
    print("==", p)
    for c in sorted(tab[p]):
        mx, av, n = tab[p][c]
        print(f"   {c:40s} max/dispatch {mx:16.0f}   mean {av:16.0f}   ({n} rows)")
