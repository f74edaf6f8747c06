import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    import tools.bench_panel as B
    for name, N, K, mode in [("fc fwd", 3072, 768, "blg"), ("proj fwd", 768, 3072, "blr"), ("proj dX", 3072, 768, "lkd"), ("fc dX", 768, 3072, "lk")]:
        B.case(name, N, K, mode)
else:
    for dbg in (0, 2, 8, 16, 18, 26):
        env = dict(os.environ, FFM_PANEL_DBG=str(dbg))
        out = subprocess.run([sys.executable, __file__, "x"], env=env, capture_output=True, text=True).stdout
        print("dbg", dbg)
        for l in out.splitlines():
            if "panel" in l:
                print("   ", l.split("|")[0][:32], "|", l.split("|")[1])
