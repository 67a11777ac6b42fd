"""Per-chain leapfrog latency of the one-wave-per-chain kernel at small D: every tree forced to max depth
(eps tiny), so all chains do identical work and throughput / resident chains = single-chain rate."""
import sys, numpy as np
sys.path.insert(0, '.')
from __graft_entry__ import load_package
pkg = load_package()
for D, C in ((30, 1024), (30, 4096), (30, 16384), (100, 4096), (1000, 1024)):
    ctx = pkg.DeviceContext(D, C, seed=1); ctx.init(); ctx.set_stepsize(1e-3)
    ctx.run_into(1, {}, allow_failure=True)
    ctx.run_into(4, {}, allow_failure=True)
    lf, ms = ctx.last_run_leapfrogs(), ctx.last_run_kernel_ms()
    print(f"D={D} chains={C}: {lf/ms*1e3:.3e} steps/s, {ms*1e3/(lf/C):.2f} us per leapfrog per chain (all {C} chains co-resident if <= {1024*8})")
