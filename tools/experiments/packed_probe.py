"""One config-4 run for profilers: C funnel chains (D = 30), a short adaptation, then T transitions; prints the timed call's
kernel ms and leapfrogs.  PH_STUCK=1: a tiny step so that every tree runs to max_depth (the straggler regime).
usage: packed_probe.py [chains] [transitions]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
pkg = load_package()
C = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ctx = pkg.DeviceContext(int(os.environ.get("PH_D", 30)), C, target=pkg.abi.TARGET_FUNNEL, seed=1)
ctx.init()
if os.environ.get("PH_STUCK"):
    ctx.set_stepsize(1e-4)
else:
    ctx.find_initial_stepsize(); ctx.run(100, da={}, fields=[])
ctx.run(T, fields=[])
ms, lf = ctx.last_run_kernel_ms(), ctx.last_run_leapfrogs()
print(f"chains {C} transitions {T} kernel_ms {ms:.3f} leapfrogs {lf} -> {lf / ms * 1e3:.4g} /s")
