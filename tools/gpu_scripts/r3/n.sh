#!/bin/bash
export PYTHONPATH=tests
O=gpurun_out/r3n; mkdir -p $O
for lib in default funnel_nolicm; do
  for ch in 4096 32768; do
    if [ $lib = default ]; then unset DHMC_LIB_PATH; else export DHMC_LIB_PATH=$PWD/tools/experiments/_v/$lib/libdhmc_amd.so; fi
    timeout -s KILL 300 python bench.py --config 4 --chains $ch --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/c4_${lib}_$ch.json
    python -c "
import json; d = json.load(open('$O/c4_${lib}_$ch.json')); print('$lib chains $ch: %.4g steps/s' % d['value'], d['tree'])" | tee -a $O/log.txt
  done
done
