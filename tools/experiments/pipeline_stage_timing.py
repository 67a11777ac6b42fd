"""Which wave of the pipeline kernel (nuts_pipeline_kernel.hpp) is the bottleneck: clocks each role spends in all and waiting
(library built by `FAM=FunnelT phase_timing.sh build`, selected through DHMC_LIB_PATH).  usage: pipeline_stage_timing.py [chains] [transitions]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["DHMC_PIPELINE"] = "1"
from __graft_entry__ import load_package
pkg = load_package()
lib = pkg.abi.lib()
C = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = pkg.DeviceContext(30, C, target=pkg.abi.TARGET_FUNNEL, seed=1)
ctx.init()
if os.environ.get("PH_STUCK"):
    ctx.set_stepsize(1e-4); ctx.run(2, fields=[])
else:
    ctx.find_initial_stepsize(); ctx.run(100, da={}, fields=[])
lib.dhmc_debug_phase.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.dhmc_debug_phase(None, 1)
ctx.run(T, fields=[]); ms = ctx.last_run_kernel_ms(); lf = ctx.last_run_leapfrogs()
ph = np.zeros(16, np.uint64)
lib.dhmc_debug_phase(ph.ctypes.data, 0)
leaves = float(ph[14]); pipes = int(ph[15])
print(f"chains {C} transitions {T} kernel_ms {ms:.3f} leapfrogs {lf} -> {lf / ms * 1e3:.4g} /s; pipelines {pipes}")
for r, name in enumerate(("A  integrator", "B1 turn statistics", "B2 proposals", "B3 visited statistic")):
    tot, wait = float(ph[4 + r]), float(ph[r])
    print(f"  {name:20s} {tot / max(leaves, 1):8.0f} clocks per leaf, of which waiting {wait / max(leaves, 1):8.0f} ({100 * wait / max(tot, 1):5.1f} %)  -> busy {(tot - wait) / max(leaves, 1):8.0f}")
print("  B2's leaf loop, clocks per leaf: record read and bookkeeping %.0f, merge cascade (logaddexp, picks) %.0f, suspension (slot store, level scalars) %.0f"
      % tuple(float(ph[8 + i]) / max(leaves, 1) for i in range(3)))
