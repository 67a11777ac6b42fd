"""Where nuts_run_kernel<StdNormalT,16,true> spends its clocks (library built by phase_timing.sh build with
-DDHMC_PHASE_TIMING and selected through DHMC_LIB_PATH).  BASELINE config 2's sampling phase: D = 1000, 4096 chains."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
pkg = load_package()
lib = pkg.abi.lib()
C = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
PH_D = int(os.environ.get("PH_D", 1000))
if os.environ.get("PH_TARGET") == "funnel":   # Neal's funnel (config 4's model), FAM=FunnelT build: a short adaptation first
    ctx = pkg.DeviceContext(PH_D, C, target=pkg.abi.TARGET_FUNNEL, seed=1)
    ctx.init(); ctx.find_initial_stepsize(); ctx.run(100, da={}, fields=[])
else:
    ctx = pkg.DeviceContext(PH_D, C, seed=1)
    ctx.init(); ctx.set_stepsize(0.3)          # unit metric, eps 0.3: depth-4 trees as after adaptation
ctx.run(20, fields=[])
lib.dhmc_debug_phase.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.dhmc_debug_phase(None, 1)
ctx.run(T, fields=[]); ms = ctx.last_run_kernel_ms(); lf = ctx.last_run_leapfrogs()
ph = np.zeros(16, np.uint64)
lib.dhmc_debug_phase(ph.ctypes.data, 0)
names = ["outside", "momentum+setup", "edge switch", "leaf", "leaf scalars", "merge vector", "merge scalar", "suspend", "end of transition", "-"]
tot = float(ph[:10].sum())
print(f"chains {C} transitions {T} kernel_ms {ms:.3f} leapfrogs {lf} ({lf / C / T:.2f} per transition) -> {lf / ms * 1e3:.3e} /s  waves {int(ph[15])}")
print(f"clocks per wave {tot / max(int(ph[15]), 1):.0f}; per leapfrog {tot / lf:.1f}")
for n, v in zip(names, ph[:10]):
    print(f"  {n:20s} {float(v) / tot * 100:6.2f} %   {float(v) / lf:8.1f} clocks per leapfrog")
