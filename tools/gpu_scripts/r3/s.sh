#!/bin/bash
export PYTHONPATH=tests
O=gpurun_out/r3s; mkdir -p $O; rm -f $O/log.txt
for parts in 1 2 3 4; do
  DHMC_DENSE_PARTS=$parts timeout -s KILL 300 python bench.py --config 3 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_c3_parts$parts.json
  python -c "
import json; d = json.load(open('$O/bench_c3_parts$parts.json')); print('parts $parts: %.4g steps/s' % d['value'], 'frac %.3f' % d['roofline']['frac'])" | tee -a $O/log.txt
done
