import sys, os
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from tools.bench_gemm import time_gemm
dt = torch.bfloat16
for mode in ["", "b", "bg", "bl", "blg", "d", "ld", "r", "br", "blr"]:
    us, tf = time_gemm(6304, 3072, 768, dt, mode)
    print(f"N3072 K768 [{mode:4s}] {us:7.1f} us {tf:7.1f} TF/s")
for mode in ["", "br", "blr", "l"]:
    us, tf = time_gemm(6304, 768, 3072, dt, mode)
    print(f"N768 K3072 [{mode:4s}] {us:7.1f} us {tf:7.1f} TF/s")
