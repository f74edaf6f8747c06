#!/bin/bash
# SQ counters of the attention kernels (tools/bench_attn.py) from one rocprofv3 --pmc pass:
#   tools/attn_pmc.sh [pattern] [bench_attn args]   (GPU box; FFM_ATTN / ATTN_SETS in the environment as for bench_attn.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
pat=${1:-attn}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/attn_pmc
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d /tmp/attn_pmc -o run -- python3 $R/tools/bench_attn.py ${@:2} > /dev/null 2>&1
f=$(find /tmp/attn_pmc -name 'run_counter_collection.csv' | head -1)
python3 - "$f" "$pat" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if sys.argv[2] not in k: continue
    k = k[22:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k, c in acc.items():
    d = len(n[k]); wc = c["SQ_WAVE_CYCLES"]; gui = c["GRBM_GUI_ACTIVE"] / 8 / d
    print(f"{k}: dispatches {d}, gui cycles/xcd {gui:.0f}, mfma_util {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * d * 1024):.3f}, "
          f"wait_any {c['SQ_WAIT_ANY'] / wc:.2f}, wait_inst {c['SQ_WAIT_INST_ANY'] / wc:.2f} (lds {c['SQ_WAIT_INST_LDS'] / wc:.2f}), active {c['SQ_ACTIVE_INST_ANY'] / wc:.2f}, "
          f"lds_idx_active/dispatch {c['SQ_LDS_IDX_ACTIVE'] / d:.0f}, lds_bank_conflict/dispatch {c['SQ_LDS_BANK_CONFLICT'] / d:.0f} "
          f"(ratio {c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1):.3f})")
PY
