#!/usr/bin/env python3
"""Timeline statistics of a rocprofv3 kernel trace of bench.py: per-stream busy time, union busy time and
idle gaps inside the timed steps (the last `steps` train steps are delimited by sgd_dev_kernel launches)."""
import csv
import re
import sys
from collections import defaultdict

path, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
rows.sort()
sgd = [i for i, r in enumerate(rows) if "sgd" in r[3]]
lo, hi = sgd[-steps - 1], sgd[-1]
win = rows[lo + 1:hi + 1]
t0, t1 = rows[lo][1], rows[hi][1]
span = (t1 - t0) / 1e3
print(f"{steps} steps span {span:.1f} us -> {span / steps:.1f} us/step, {len(win) / steps:.0f} kernels/step")
byq = defaultdict(float)
for s, e, q, n in win:
    byq[q] += (e - s) / 1e3
for q, v in sorted(byq.items(), key=lambda x: -x[1]):
    print(f"  queue {q}: busy {v / steps:8.1f} us/step")
# union busy
ev = sorted([(s, 1) for s, e, q, n in win] + [(e, -1) for s, e, q, n in win])
busy, depth, last = 0, 0, None
for t, d in ev:
    if depth > 0:
        busy += t - last
    depth += d
    last = t
print(f"  union busy {busy / 1e3 / steps:.1f} us/step, idle {(t1 - t0 - busy) / 1e3 / steps:.1f} us/step")
# main queue = busiest; gaps between consecutive kernels on it
mq = max(byq, key=byq.get)
mk = [r for r in win if r[2] == mq]
gaps = [(mk[i + 1][0] - mk[i][1]) / 1e3 for i in range(len(mk) - 1)]
print(f"  main queue {mq}: {len(mk) / steps:.0f} kernels/step, sum of gaps {sum(g for g in gaps if g > 0) / steps:.1f} us/step, "
      f"median gap {sorted(gaps)[len(gaps) // 2]:.2f} us")
agg = defaultdict(lambda: [0, 0.0])
for s, e, q, n in win:
    n = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+", "", n)[:70]
    agg[(q == mq, n)][0] += 1
    agg[(q == mq, n)][1] += (e - s) / 1e3
for (m, n), (c, v) in sorted(agg.items(), key=lambda x: -x[1][1])[:28]:
    print(f"  {'main' if m else 'side'} {c / steps:6.1f}x {v / steps:8.1f} us/step  {v / c:7.1f} avg  {n}")
