"""The reference's own use: a handful of chains.  D-dimensional standard normal (BASELINE configs[0]: D = 100, 4 chains), a short
adaptation, then 1000 transitions in one call: kernel ms, leapfrogs, µs per leapfrog of a chain.  Engine by DHMC_PIPELINE / DHMC_PACKED.
usage: few_chain_latency.py [D] [chains] [name]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
pkg = load_package()
D = int(sys.argv[1]) if len(sys.argv) > 1 else 100
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4
name = sys.argv[3] if len(sys.argv) > 3 else ""
tgt = pkg.abi.TARGET_FUNNEL if os.environ.get("PH_FUNNEL") else pkg.abi.TARGET_STD_NORMAL
ctx = pkg.DeviceContext(D, C, target=tgt, seed=1)
ctx.init(); ctx.find_initial_stepsize(); ctx.run(150, da={}, fields=[])
ctx.run(1000, fields=[])
ms, lf = ctx.last_run_kernel_ms(), ctx.last_run_leapfrogs()
print(f"{name:10s} D {D:4d} chains {C:4d} kernel_ms {ms:9.3f} leapfrogs {lf:9d} -> {ms * 1e3 / (lf / C):.3f} us per leapfrog of a chain, {lf / ms * 1e3:.4g} /s", flush=True)
