#!/usr/bin/env python3
"""Concurrency in a rocprofv3 --kernel-trace CSV: over the last `frac` of the trace (the timed graph replays), the wall span, the sum of
kernel durations, the time with >= 2 kernels in flight, and which kernel pairs overlap most.  usage: overlap_stats.py <kernel_trace.csv> [frac]"""
import collections, csv, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kname import canon
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), canon(r["Kernel_Name"]).split("<")[0], r.get("Queue_Id", "?")) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
t_lo = rows[0][0] + (rows[-1][1] - rows[0][0]) * (1 - frac)
rows = [r for r in rows if r[0] >= t_lo]
span = max(r[1] for r in rows) - rows[0][0]
busy = sum(e - s for s, e, _, _ in rows)
ev = sorted([(s, 1, i) for i, (s, e, _, _) in enumerate(rows)] + [(e, -1, i) for i, (s, e, _, _) in enumerate(rows)])
active, last, t2, t1, t0 = set(), ev[0][0], 0, 0, 0
pairs = collections.Counter()
for t, d, i in ev:
    dt = t - last
    if len(active) >= 2:
        t2 += dt
        names = sorted(rows[j][2] for j in list(active)[:2])
        pairs[tuple(names)] += dt
    elif len(active) == 1:
        t1 += dt
    else:
        t0 += dt
    last = t
    if d > 0: active.add(i)
    else: active.discard(i)
print(f"kernels {len(rows)}  queues {sorted(set(r[3] for r in rows))}  span {span / 1e6:.2f} ms  sum of durations {busy / 1e6:.2f} ms  >=2 in flight {t2 / 1e6:.2f} ms  exactly 1 {t1 / 1e6:.2f} ms  idle {t0 / 1e6:.2f} ms")
for (a, b), dt in pairs.most_common(12):
    print(f"  {dt / 1e6:7.2f} ms  {a}  ||  {b}")
