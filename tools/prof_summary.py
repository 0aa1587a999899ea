#!/usr/bin/env python3
"""Summarise a rocprofv3 results database (kernel trace / PMC) as text.

    python tools/prof_summary.py gpurun_out/prof/<host>/<pid>_results.db [--pmc]

rocprofv3 7.2 writes a rocpd SQLite database by default; this prints the `--stats` view
(per-kernel calls / total / average / share) and, with --pmc, per-kernel counter averages, so that
the summary can be committed under profiles/.
"""
import sqlite3
import sys


def main():
    path = sys.argv[1]
    db = sqlite3.connect(path)
    cur = db.cursor()
    print("# source: %s" % path)
    print("%-90s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in cur.execute(
            "select name, total_calls, total_duration, average, percentage from top_kernels"):
        short = name if len(name) <= 88 else name[:85] + "..."
        print("%-90s %8d %14.3f %12.3f %7.2f" % (short, calls, total, avg, pct))
    if "--pmc" in sys.argv:
        print()
        print("%-60s %-28s %10s %18s" % ("kernel", "counter", "dispatches", "avg_value"))
        q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
             "group by kernel_name, counter_name order by kernel_name, counter_name")
        for name, cname, n, avg in cur.execute(q):
            short = name if len(name) <= 58 else name[:55] + "..."
            print("%-60s %-28s %10d %18.1f" % (short, cname, n, avg))


if __name__ == "__main__":
    main()
