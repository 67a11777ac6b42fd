import sys, numpy as np
sys.path.insert(0,'.')
from __graft_entry__ import load_package
pkg=load_package()
ctx=pkg.DeviceContext(1000,4096,seed=1); ctx.init(); ctx.set_stepsize(0.27)
ctx.run_into(10,{},allow_failure=True)
tot=0; ms=0
for _ in range(3):
    ctx.run_into(20,{},allow_failure=True); tot+=ctx.last_run_leapfrogs(); ms+=ctx.last_run_kernel_ms()
print("steps/s", tot/ms*1e3, "leapfrogs/transition", tot/(3*20*4096))
