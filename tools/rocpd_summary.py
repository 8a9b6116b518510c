#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) run: per-kernel stats and PMC sums.

    python tools/rocpd_summary.py <results.db> [--pmc]  > profiles/<name>.txt
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    cur = con.cursor()
    print("# rocprofv3 --kernel-trace summary of %s" % db)
    print("%-64s %8s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"))
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                       "group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    for name, n, s, a, mn, mx in rows:
        print("%-64s %8d %14.3f %12.2f %12.2f %12.2f %6.2f%%" % (name[:64], n, s / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
    if "--pmc" in sys.argv:
        cols = [r[1] for r in cur.execute("pragma table_info('pmc_events')")]
        print("\n# PMC (per kernel: sum and mean per dispatch); columns of pmc_events: %s" % cols)
        q = ("select k.name, p.counter_name, count(*), sum(p.counter_value), avg(p.counter_value) from pmc_events p "
             "join kernels k on k.dispatch_id = p.dispatch_id group by k.name, p.counter_name order by sum(p.counter_value) desc")
        try:
            for name, cname, n, s, a in cur.execute(q):
                print("%-64s %-14s n=%6d sum=%16.1f mean=%14.2f" % (name[:64], cname, n, s, a))
        except sqlite3.Error as e:
            print("pmc query failed:", e)


if __name__ == "__main__":
    main()
