import sys, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from __graft_entry__ import load_package
import oracle_lib as ol
pkg=load_package()
D=1000; rho=0.5
sig=np.logspace(-1,1,D)
Pc=np.zeros(D)+(1+rho**2)/(1-rho**2); Pc[0]=Pc[-1]=1/(1-rho**2)
diag=Pc/sig**2; off=np.zeros(D); off[:D-1]=-rho/(1-rho**2)/(sig[:-1]*sig[1:])
idx=np.arange(D); Sigma=np.outer(sig,sig)*rho**np.abs(idx[:,None]-idx[None,:])
import os
CH=[int(x) for x in os.environ.get('DENSE_CHAINS','256,1024,4096').split(',')]
NT=int(os.environ.get('DENSE_TRANSITIONS','5'))
for C in CH:
    dev=pkg.DeviceContext(D,C,metric=ol.METRIC_DENSE,target=ol.TARGET_TRIDIAG_NORMAL,target_params=np.concatenate([diag,off]),seed=3)
    t0=time.time(); dev.set_metric_dense(Sigma); t1=time.time()
    dev.init(np.random.default_rng(5).normal(size=(C,D))*sig); dev.set_stepsize(0.4)
    dev.run(2,fields=[]) 
    dev.run(NT,fields=["steps"]); ms=dev.last_run_kernel_ms(); lf=dev.last_run_leapfrogs(); rd=dev.last_run_rounds()
    gemm_flops=rd*2*2.0*C*1024*1024
    print(C,"chains: set_metric_dense",round(t1-t0,2),"s;",NT,"transitions",round(ms,1),"ms",lf,"leapfrogs",rd,"rounds ->",lf/ms*1e3,"steps/s; big-GEMM flops/total time =",gemm_flops/ms/1e9,"TFLOP/s")
