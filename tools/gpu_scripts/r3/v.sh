#!/bin/bash
export PYTHONPATH=tests
O=gpurun_out/r3v; mkdir -p $O; rm -f $O/log.txt
for sp in 0 50 60 70 80; do
  DHMC_DENSE_CU_SPLIT=$sp timeout -s KILL 300 python bench.py --config 3 --steps 3 --warmup 1 2>$O/err_$sp.txt | tail -1 > $O/bench_c3_split$sp.json
  python -c "
import json; d = json.load(open('$O/bench_c3_split$sp.json')); print('split $sp: %.4g steps/s' % d['value'], 'frac %.3f' % d['roofline']['frac'], d['tree']['scaled_draw_var'])" | tee -a $O/log.txt
done
for sp in 60 70; do
  DHMC_DENSE_PARTS=4 DHMC_DENSE_CU_SPLIT=$sp timeout -s KILL 300 python bench.py --config 3 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_c3_split${sp}_p4.json
  python -c "
import json; d = json.load(open('$O/bench_c3_split${sp}_p4.json')); print('split $sp parts 4: %.4g steps/s' % d['value'])" | tee -a $O/log.txt
done
