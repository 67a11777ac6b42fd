"""Packed vs wave-per-chain engine on chains of UNIFORM work: a standard normal of D coordinates (every tree has the same
depth), C chains, T transitions after a short adaptation.  usage: packed_uniform_probe.py D C T"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
pkg = load_package()
D, C, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
for name, env in (("wave", dict(DHMC_PACKED="0")), ("packed cpl2", dict(DHMC_PACKED="1", DHMC_PK="cpl=2")), ("packed cpl4", dict(DHMC_PACKED="1", DHMC_PK="cpl=4"))):
    os.environ.update(env)
    ctx = pkg.DeviceContext(D, C, seed=1)
    ctx.init(); ctx.find_initial_stepsize(); ctx.run(60, da={}, fields=[])
    ctx.run(T, fields=[])
    ms, lf = ctx.last_run_kernel_ms(), ctx.last_run_leapfrogs()
    print(f"D {D} chains {C} transitions {T} {name:12s} kernel_ms {ms:9.3f} leapfrogs {lf} -> {lf / ms * 1e3:.4g} /s")
    ctx.close()
