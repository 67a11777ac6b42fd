#!/bin/bash
# round 6, second session: rounds_k3b_kernel (211 VGPRs: two waves per SIMD) forced to 3 / 4 waves per SIMD by launch bounds (167 VGPRs + 108 B of
# scratch / 128 + 252 B): same-box runs of config 3, and the dense parity tests on the faster one
O=gpurun_out/r6bs; mkdir -p $O
for i in 1 2; do
 for v in lib k3b_occ3 k3b_occ4; do
   if [ $v = lib ]; then unset DHMC_LIB_PATH; else export DHMC_LIB_PATH=$PWD/tools/experiments/_v/$v/libdhmc_amd.so; fi
   r=$(timeout -s KILL 300 python bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline 2>$O/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g' % d['value'])")
   echo "c3 $v: $r" | tee -a $O/ab.txt
 done
done
