"""Summarise rocprofv3 (rocpd sqlite) outputs into the text files committed under profiles/.
usage: python tools/prof_summary.py <trace.db> [<pmc.db> ...] > profiles/rNN_summary.txt"""
import sqlite3
import sys


def kernel_stats(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"== kernel trace: {path}   (tools/gpu_pass.sh traces the SERIAL frame, PNR_OVERLAP=0: every MLP dispatch a whole-device launch)")
    print(f"{'kernel':72s} {'calls':>6s} {'total_ms':>10s} {'pct':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s}")
    for r in rows[:14]:
        print(f"{r[0][:72]:72s} {r[1]:6d} {r[2] / 1e6:10.3f} {100 * r[2] / tot:6.2f} {r[3] / 1e3:10.1f} {r[4] / 1e3:10.1f} {r[5] / 1e3:10.1f}")
    # the bench's dominant launch, selected BY NAME: the MLP kernel with the largest total time (k_mlp_tt_* / k_mlp_pp<..> /
    # k_mlp_fused<..>; a single comparison launch of another form no longer leaks into the average or lends its launch shape --
    # VERDICT r5 weak #8); its fine-level chunk launches are the dispatches of that name above 70 % of its longest
    mlp = [r for r in rows if any(k in r[0] for k in ("k_mlp_tt", "k_mlp_pp", "k_mlp_fused"))]
    if mlp:
        name = mlp[0][0]
        big = cur.execute("select (end-start)/1e3, grid_x, workgroup_x, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size "
                          "from kernels where name = ? order by 1 desc", (name,)).fetchall()
        top = [b for b in big if b[0] > 0.7 * big[0][0]]
        print(f"dominant kernel {name[:60]}: fine-level chunk launches (>70% of its longest): n={len(top)} "
              f"avg={sum(b[0] for b in top) / len(top):.1f} us grid={top[0][1]} wg={top[0][2]} vgpr={top[0][3]} agpr={top[0][4]} "
              f"sgpr={top[0][5]} lds={top[0][6]} scratch={top[0][7]}")
    print()


def pmc(path):
    cur = sqlite3.connect(path).cursor()
    print(f"== PMC: {path}")
    rows = cur.execute("select kernel_name, counter_name, count(*), sum(value), max(value) from counters_collection "
                       "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
    for r in rows:
        if any(k in r[0] for k in ("k_mlp", "k_composite", "k_sample_pdf")):
            print(f"{r[0][:60]:60s} {r[1]:28s} dispatches={r[2]:4d} sum={r[3]:.6g} max_per_dispatch={r[4]:.6g}")
    print()


if __name__ == "__main__":
    kernel_stats(sys.argv[1])
    for p in sys.argv[2:]:
        pmc(p)
