#!/usr/bin/env python3
"""HBM-side traffic of the GEMM kernels from two rocprofv3 PMC passes (MI355X_MICROARCH.md, HBM section):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out/fetch -o run --output-format csv -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d out/write -o run --output-format csv -- python3 bench.py ...
    python3 tools/pmc_traffic.py out/fetch/run_counter_collection.csv out/write/run_counter_collection.csv > profiles/rNN_traffic.json

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch; on gfx950 FETCH_SIZE tallies a 128-byte request as 64
bytes for wide coalesced reads, so it is doubled (the guide's correction).  The vision-tower GEMM launches are the
dispatches of gemm_panel_kernel / gemm_nt_kernel with >= 200 workgroups (the text tower's use 12..48)."""
import csv
import json
import re
import sys
from collections import defaultdict


MIN_WGS = int(sys.argv[3]) if len(sys.argv) > 3 else 200      # (RN50, --config c5: 40 - its layer3 / layer4 products launch 52-208 tiles)


def load(path, counter):
    per = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        name = r["Kernel_Name"]
        if "gemm_panel_kernel" not in name and "gemm_nt_kernel" not in name and "conv_narrow_kernel" not in name:
            continue
        wgs = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1) if r.get("Grid_Size") else 0
        if wgs < MIN_WGS:
            continue
        short = re.sub(r"\(anonymous namespace\)::|ffm_panel::|void |\(ffm_gemm_args\)", "", name)[:60]
        per[short][0] += 1
        per[short][1] += float(r["Counter_Value"])
    return per


if __name__ == "__main__":
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    out, tot_n, tot_b = {}, 0, 0.0
    for k in sorted(set(fetch) | set(write)):
        n = fetch.get(k, [0, 0])[0] or write.get(k, [0, 0])[0]
        f = 2.0 * 1024.0 * fetch.get(k, [0, 0.0])[1] / max(fetch.get(k, [1, 0])[0], 1)       # gfx950: x2
        w = 1024.0 * write.get(k, [0, 0.0])[1] / max(write.get(k, [1, 0])[0], 1)
        out[k] = {"launches": n, "fetch_bytes_per_launch": f, "write_bytes_per_launch": w}
        tot_n += n
        tot_b += n * (f + w)
    print(json.dumps({"unit": "bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE)", "vision_gemm_launches": tot_n,
                      "traffic_bytes_per_launch": tot_b / max(tot_n, 1), "kernels": out}, indent=1))
