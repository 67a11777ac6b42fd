#!/bin/bash
# same-box A/B of two library builds:  n_gemm_ab.sh <variant dir under tools/experiments/_v> "<configs>"
O=gpurun_out/r6bn_$1; mkdir -p $O
for c in ${2:-3 5}; do
 for i in 1 2 3; do
  for v in lib $1; do
   if [ $v = lib ]; then unset DHMC_LIB_PATH; else export DHMC_LIB_PATH=$PWD/tools/experiments/_v/$v/libdhmc_amd.so; fi
   r=$(timeout -s KILL 300 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline 2>$O/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g' % d['value'])")
   echo "c$c $v: $r" | tee -a $O/ab.txt
  done
 done
done
