#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) run: per-kernel stats and PMC sums.

    python tools/rocpd_summary.py <results.db> [--pmc] [--busy SUBSTRING ...]  > profiles/<name>.txt

--busy NAME: for the kernels whose name contains NAME, the time during which AT LEAST ONE of their dispatches was running (union of the
[start, end] intervals) beside the sum of the dispatches' own durations -- with several tick streams the dispatches of a kernel overlap
each other, and bytes / busy time is the bandwidth the kernel achieved while it ran (bench.py's roofline uses the same definition
from HIP events: roofline.kernel_ms_per_step x steps for the timed region, all_launches.busy_ms for the whole process).
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
    names = [sys.argv[i + 1] for i, a in enumerate(sys.argv[:-1]) if a == "--busy"]
    if names:
        cols = [r[1] for r in cur.execute("pragma table_info('kernels')")]
        sc = next((c for c in ("start", "start_time", "begin") if c in cols), None)
        ec = next((c for c in ("end", "end_time", "stop") if c in cols), None)
        print("\n# busy time (union of the dispatch intervals) per kernel-name substring; columns of kernels: %s" % cols)
        for nm in names:
            if sc is None or ec is None:
                print("%-40s no start/end columns in this rocpd version" % nm)
                continue
            iv = sorted(cur.execute("select %s, %s from kernels where name like ?" % (sc, ec), ("%" + nm + "%",)).fetchall())
            busy, lo, hi, tot = 0, None, None, 0
            for a, b in iv:
                tot += b - a
                if hi is None or a > hi:
                    if hi is not None:
                        busy += hi - lo
                    lo, hi = a, b
                else:
                    hi = max(hi, b)
            if hi is not None:
                busy += hi - lo
            print("%-40s dispatches=%6d  sum_of_durations_ms=%12.3f  busy_ms=%12.3f  in_flight=%6.3f  avg_us=%10.2f" % (
                nm, len(iv), tot / 1e6, busy / 1e6, (tot / busy) if busy else 0.0, (tot / len(iv) / 1e3) if iv else 0.0))
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
