#!/usr/bin/env python3
"""Kernel timeline of a rocprofv3 --kernel-trace database: start / end of the fused step kernels per queue,
their durations, the gap to the previous kernel of the same queue, and how much of each kernel overlapped
kernels of other queues.

    python tools/timeline.py <results.db> [first [count]]
"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tabs else None
if view is None:
    print("tables:", tabs); sys.exit(1)
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % view)]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(cur.execute("select name, start, end, %s from %s order by start" % (qcol or "0", view)))
rows = [r for r in rows if "rollout" in r[0]]
first = int(sys.argv[2]) if len(sys.argv) > 2 else max(0, len(rows) // 2)
count = int(sys.argv[3]) if len(sys.argv) > 3 else 24
t0 = rows[first][1]
last_end = {}
print("columns:", cols)
print("%4s %6s %10s %10s %8s %8s" % ("idx", "queue", "start_us", "end_us", "dur_us", "gap_us"))
import collections
durs, gaps = collections.defaultdict(list), collections.defaultdict(list)
for i, (name, s, e, q) in enumerate(rows):
    gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
    last_end[q] = e
    if i >= first and i < first + count:
        print("%4d %6s %10.2f %10.2f %8.2f %8.2f" % (i, q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, gap))
    if i > 10:
        durs[q].append((e - s) / 1e3); gaps[q].append(gap)
for q in durs:
    d, g = sorted(durs[q]), sorted(gaps[q])
    print("queue %s: %d kernels, duration median %.2f us (p10 %.2f p90 %.2f), gap median %.2f us (p10 %.2f p90 %.2f)" % (
        q, len(d), d[len(d) // 2], d[len(d) // 10], d[9 * len(d) // 10], g[len(g) // 2], g[len(g) // 10], g[9 * len(g) // 10]))
