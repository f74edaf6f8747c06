#!/usr/bin/env python3
"""Is the text tower on the critical path?  From a rocprofv3 kernel trace of bench.py (tools/prof.sh): per step, the end of the
text forward (text_tail_norm_kernel, side queue) against the start of head_fwd_kernel (main queue) and the end of the last
vision-forward kernel before it; and the end of the text backward (text_ctx_grad_kernel) against the start of the SGD kernel."""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
rows.sort()
heads = [i for i, r in enumerate(rows) if "head_fwd_kernel" in r[3]]
out = []
for hi in heads[-8:]:
    hs, hq = rows[hi][0], rows[hi][2]
    tt = max((r[1] for r in rows[:hi] if "text_tail_norm" in r[3]), default=None)
    prev_main = max((r[1] for r in rows[:hi] if r[2] == hq), default=None)
    sg = next((r for r in rows[hi:] if "sgd" in r[3]), None)
    cg = max((r[1] for r in rows[hi:rows.index(sg)] if "text_ctx_grad" in r[3]), default=None) if sg else None
    lastmain = max((r[1] for r in rows[hi:rows.index(sg)] if r[2] == hq), default=None) if sg else None
    out.append(((hs - tt) / 1e3, (hs - prev_main) / 1e3, (sg[0] - cg) / 1e3 if cg else None, (sg[0] - lastmain) / 1e3 if lastmain else None))
print("per step (us): head_fwd start - text forward end | head_fwd start - previous main-queue kernel end | sgd start - text backward end | sgd start - previous main-queue kernel end")
for o in out:
    print("  " + " | ".join("%8.1f" % v if v is not None else "     n/a" for v in o))
