#!/usr/bin/env python3
"""Condense rocprofv3 result databases (rocpd sqlite) into the small text summaries kept under profiles/.

usage: rocprof_summary.py trace <results.db>          per-kernel stats (calls, total, average, %)
       rocprof_summary.py pmc <results.db> [filter]   per-kernel mean of every collected counter
"""
import sqlite3
import sys


def short(name, n=110):
    name = name.replace("void ", "")
    return name if len(name) <= n else name[:n - 3] + "..."


def trace(db):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print(f"# rocprofv3 --kernel-trace --stats   ({db})")
    print(f"{'kernel':112s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'%':>7s}")
    for name, calls, total, avg, pct in rows:
        print(f"{short(name):112s} {calls:7d} {total:12.1f} {avg:10.3f} {pct:7.2f}")
    # register / LDS / scratch footprint of our kernels as the runtime saw them
    q = ("select name, max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
         "max(grid_x), max(workgroup_x), min(duration), max(duration) from kernels where name like '%k_%' group by name")
    print("\n# dispatch footprint (vgpr, agpr, sgpr, lds B, scratch B, grid, block, min ns, max ns)")
    for r in cur.execute(q):
        print(f"{short(r[0], 70):72s}", *r[1:])


def pmc(db, flt="k_step"):
    cur = sqlite3.connect(db).cursor()
    q = ("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection "
         "group by kernel_name, counter_name")
    print(f"# rocprofv3 --pmc   ({db})   mean per dispatch")
    print(f"{'kernel':72s} {'counter':>14s} {'n':>6s} {'mean':>14s} {'min':>14s} {'max':>14s}")
    for name, ctr, n, mean, lo, hi in cur.execute(q):
        if flt in name:
            print(f"{short(name, 70):72s} {ctr:>14s} {n:6d} {mean:14.3f} {lo:14.3f} {hi:14.3f}")


if __name__ == "__main__":
    if sys.argv[1] == "trace":
        trace(sys.argv[2])
    else:
        pmc(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "k_")
