#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) as a per-kernel stats table.
Usage: tools/rocpd_stats.py results.db > profiles/<name>.txt"""
import sqlite3
import sys


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = ("select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), "
         "max(d.end-d.start), max(d.workgroup_size_x), max(d.grid_size_x) from %s d join %s s "
         "on d.kernel_id=s.id group by s.kernel_name order by 3 desc" % (disp, sym))
    rows = list(cur.execute(q))
    total = sum(r[2] for r in rows) or 1
    print("%-90s %6s %14s %12s %12s %12s %6s %10s %6s" % ("Name", "Calls", "TotalDur(ns)", "Avg(ns)",
                                                          "Min(ns)", "Max(ns)", "WG", "Grid", "Pct"))
    for r in rows:
        print("%-90s %6d %14d %12.1f %12d %12d %6d %10d %6.2f" % (r[0][:90], r[1], r[2], r[3], r[4],
                                                                  r[5], r[6], r[7], 100.0 * r[2] / total))


if __name__ == "__main__":
    main(sys.argv[1])
