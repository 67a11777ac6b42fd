"""Where nuts_run_packed_kernel<funnel, 8> spends its clocks (library built by `FAM=FunnelT phase_timing.sh build`, selected
through DHMC_LIB_PATH): BASELINE config 4's chains after a short adaptation.  usage: packed_phase_timing.py [chains] [transitions]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
pkg = load_package()
lib = pkg.abi.lib()
C = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
D = int(os.environ.get("PH_D", 30))
STUCK = os.environ.get("PH_STUCK")          # every tree runs to max_depth (a tiny step): all chains of a wave busy on every trip, lane 0's clocks are the wave's
ctx = pkg.DeviceContext(D, C, target=pkg.abi.TARGET_FUNNEL, seed=1)
if STUCK:
    ctx.init(); ctx.set_stepsize(1e-4); ctx.run(2, fields=[])
else:
    ctx.init(); ctx.find_initial_stepsize(); ctx.run(100, da={}, fields=[])
    ctx.run(20, fields=[])
lib.dhmc_debug_phase.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.dhmc_debug_phase(None, 1)
ctx.run(T, fields=[]); ms = ctx.last_run_kernel_ms(); lf = ctx.last_run_leapfrogs()
ph = np.zeros(16, np.uint64)
lib.dhmc_debug_phase(ph.ctypes.data, 0)
names = ["tail", "gate: end+start (+ idle)", "doubling start", "leaf", "leaf scalars", "merge vector", "merge scalar", "suspend", "unwind / end of trip", "-"]
tot = float(ph[:10].sum()); trips = float(ph[14]); waves = int(ph[15])
print(f"chains {C} transitions {T} kernel_ms {ms:.3f} leapfrogs {lf} ({lf / C / T:.2f} per transition) -> {lf / ms * 1e3:.3e} /s  waves {waves}")
print(f"trips per wave {trips / max(waves, 1):.0f} (leapfrogs per trip and wave {lf / max(trips, 1):.2f} of {C / max(waves, 1):.0f}); clocks per trip {tot / max(trips, 1):.0f}")
for n, v in zip(names, ph[:10]):
    print(f"  {n:22s} {float(v) / tot * 100:6.2f} %   {float(v) / max(trips, 1):8.1f} clocks per trip")
