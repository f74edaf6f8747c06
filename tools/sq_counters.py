#!/usr/bin/env python3
"""SQ counters per kernel from a rocprofv3 --pmc pass (tools/pmc.sh <name> -> gpurun_out/<name>_sq.csv):
    python3 tools/sq_counters.py gpurun_out/r02_sq.csv > profiles/r02_sq_counters.json
mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); the wave fractions are of SQ_WAVE_CYCLES
(WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ 1: parked at a wait / barrier, stalled at issue, issuing)."""
import csv
import json
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(anonymous namespace\)::|ffm_panel::|void |\(ffm_gemm_args\)", "", r["Kernel_Name"])[:64]
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[name].add(r["Dispatch_Id"])
out = {}
for k, c in acc.items():
    n = len(cnt[k])
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    if not gui or c.get("SQ_WAVE_CYCLES", 0) == 0:
        continue
    wc = c["SQ_WAVE_CYCLES"]
    out[k] = {"dispatches": n, "gui_cycles_per_xcd": gui / 8 / n,
              "mfma_util": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8 * 1024), 4),
              "wave_wait_any": round(c.get("SQ_WAIT_ANY", 0.0) / wc, 3), "wave_wait_inst": round(c.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3),
              "wave_active": round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 3)}
top = dict(sorted(out.items(), key=lambda kv: -kv[1]["gui_cycles_per_xcd"] * kv[1]["dispatches"])[:24])
print(json.dumps({"source": "tools/pmc.sh (rocprofv3 --pmc SQ_* GRBM_GUI_ACTIVE --kernel-trace -- bench.py --serial)",
                  "kernels": top}, indent=1))
