import sys, time, numpy as np
sys.path.insert(0,'.'); 
from __graft_entry__ import load_package
pkg=load_package()
D,C,T=1000,4096,20
ctx=pkg.DeviceContext(D,C,seed=1); ctx.init(); ctx.find_initial_stepsize(); ctx.run(40,da={},fields=[])
arrs={k:np.zeros((C,T,D) if k=="draws" else (C,T),dt) for k,dt in (("draws",np.float64),("steps",np.int64),("depth",np.int32),("acceptance_rate",np.float64),("logdensities",np.float64))}
ctx.run_into(T,arrs)
t0=time.perf_counter(); lf=0
for _ in range(3):
    ctx.run_into(T,arrs); lf+=ctx.last_run_leapfrogs()
dt=time.perf_counter()-t0
print("host-buffer (PCIe-inclusive, pageable numpy) rate:", lf/dt, "steps/s;", dt/3*1e3, "ms per 20-transition sweep; kernel", ctx.last_run_kernel_ms(),"ms")
