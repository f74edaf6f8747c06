#!/usr/bin/env python3
"""Registers / scratch / LDS of every kernel of one csrc file as hipcc compiles it for the product build (build container):
    tools/regs.py gemm_skinny.hip [name filter] [extra hipcc flags ...]
Prints one line per kernel whose mangled name contains the filter (all kernels that spill are always printed)."""
import os, re, subprocess, sys, tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, "fairfedmed_amd", "csrc")
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
extra = [a for a in sys.argv[2:] if a.startswith("-")]
sys.path.insert(0, root)
from fairfedmed_amd import build as B

with tempfile.TemporaryDirectory() as d:
    cmd = ["hipcc"] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + extra + ["-I", csrc, "-include", B.GEN_MAIN, "--save-temps", "-c",
                                                                     os.path.join(csrc, src), "-o", os.path.join(d, "x.o")]
    r = subprocess.run(cmd, cwd=d, capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-4000:])
    asm = [f for f in os.listdir(d) if f.endswith("gfx950.s")][0]
    s = open(os.path.join(d, asm)).read()
blk = s[s.index("amdhsa.kernels"):]
for e in blk.split("  - .agpr_count:")[1:]:
    g = lambda k: re.search(r"\." + k + r":\s+(\S+)", e).group(1)
    name, scratch = g("name"), int(g("private_segment_fixed_size"))
    if scratch or (flt and flt in name) or not flt:
        print("%-90s agpr %3s vgpr %3s sgpr %3s scratch %4d lds %6s" % (name[:90], e.split()[0], g("vgpr_count"), g("sgpr_count"), scratch,
                                                                       g("group_segment_fixed_size")))
