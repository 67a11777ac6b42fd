"""Throughput of the per-draw engines on funnel chains when no chain can hold the launch open: (a) stuck chains (step 1e-4: every tree
1023 leapfrogs, all chains in step), (b) adapted funnel chains with the tree depth limited to PH_DEPTH (default 5: at most 31
leapfrogs per transition, so the heavy tail of the tree sizes is cut and the launch is bound by throughput, with the chains out of
step as in the real run).  usage: packed_bulk_probe.py [chains] ; engine by DHMC_PACKED / DHMC_PIPELINE / DHMC_PK_* as usual"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
pkg = load_package()
C = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
name = sys.argv[2] if len(sys.argv) > 2 else ""
ctx = pkg.DeviceContext(30, C, target=pkg.abi.TARGET_FUNNEL, seed=1)
ctx.init(); ctx.set_stepsize(1e-4)
ctx.run(1, fields=[])
ctx.run(4, fields=[])
ms, lf = ctx.last_run_kernel_ms(), ctx.last_run_leapfrogs()
print(f"{name:28s} stuck   chains {C} kernel_ms {ms:9.3f} leapfrogs {lf:12d} -> {lf / ms * 1e3:.4g} /s", flush=True)
depth = int(os.environ.get("PH_DEPTH", 5))
ctx = pkg.DeviceContext(30, C, target=pkg.abi.TARGET_FUNNEL, seed=1, max_depth=depth)
ctx.init(); ctx.find_initial_stepsize(); ctx.run(100, da={}, fields=[])
ctx.run(200, fields=[])
ctx.run(400, fields=[])
ms, lf = ctx.last_run_kernel_ms(), ctx.last_run_leapfrogs()
print(f"{name:28s} depth<={depth} chains {C} kernel_ms {ms:9.3f} leapfrogs {lf:12d} -> {lf / ms * 1e3:.4g} /s", flush=True)
