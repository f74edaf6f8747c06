"""Build libffm_hip.so (gfx950) in-tree with hipcc.

    python -m fairfedmed_amd.build [--force]

hipcc cross-compiles without a GPU.  Objects are rebuilt only when their source
(or a header) is newer, so repeated calls cost nothing.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(CSRC, "libffm_hip.so")
SOURCES = ["gemm.hip", "gemm_panel.hip", "gemm_panel_rk.hip", "gemm_panel_rk2.hip", "gemm_panel_rk3.hip", "gemm_skinny.hip", "rowwise.hip", "lora.hip", "head.hip", "head_ot.hip", "optim.hip", "attention.hip", "slice3d.hip", "conv.hip", "evalmetrics.hip", "evalsort.hip", "transport.hip", "text.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_panel.h"), os.path.join(CSRC, "gemm_panel_impl.h"), os.path.join(ROOT, "include", "ffm_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the FairLoRA HIP library cannot be built")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, stamps: bool = False) -> str:
    """stamps=True: the diagnostic twin libffm_hip_stamps.so whose panel GEMM writes in-kernel phase time stamps
    (tools/panel_stamps.py); every other object is shared with the product build."""
    hipcc = _hipcc()
    objs, jobs = [], []
    lib = LIB.replace(".so", "_stamps.so") if stamps else LIB
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        special = stamps and (src.startswith("gemm_panel") or src == "lora.hip")
        o = os.path.join(CSRC, src.replace(".hip", ".stamps.o" if special else ".o"))
        objs.append(o)
        if force or _stale(o, [s] + HEADERS) or (special and os.environ.get("FFM_STAMPS_DEFS")):
            extra = ["-DFFM_PANEL_STAMPS"] + os.environ.get("FFM_STAMPS_DEFS", "").split() if special else []
            jobs.append([hipcc] + FLAGS + extra + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{r.stdout}\n{r.stderr}")
        return r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if force or jobs or _stale(lib, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, stamps="--stamps" in sys.argv))
