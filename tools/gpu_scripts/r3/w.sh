#!/bin/bash
export PYTHONPATH=tests
O=gpurun_out/r3w; mkdir -p $O; rm -f $O/log.txt
for pad in 0 8 16 26 40; do
  DHMC_K3B_LDS_PAD=$pad timeout -s KILL 300 python bench.py --config 3 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_c3_pad$pad.json
  python -c "
import json; d = json.load(open('$O/bench_c3_pad$pad.json')); print('pad $pad KB: %.4g steps/s' % d['value'], 'frac %.3f' % d['roofline']['frac'])" | tee -a $O/log.txt
done
for pad in 16 26; do
  DHMC_DENSE_PARTS=4 DHMC_K3B_LDS_PAD=$pad timeout -s KILL 300 python bench.py --config 3 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_c3_pad${pad}_p4.json
  python -c "
import json; d = json.load(open('$O/bench_c3_pad${pad}_p4.json')); print('pad $pad KB, 4 parts: %.4g steps/s' % d['value'])" | tee -a $O/log.txt
done
