#!/bin/bash
set -u
O=gpurun_out/r3c; mkdir -p $O
export PYTHONPATH=tests
timeout -s KILL 900 python -m pytest tests/test_gpu_tolerance.py -m gpu -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee $O/log.txt
tail -30 $O/pytest.txt | tee -a $O/log.txt
for parts in 1 2 3 4; do
  DHMC_DENSE_PARTS=$parts timeout -s KILL 300 python bench.py --config 3 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_c3_parts$parts.json
  python -c "
import json; d = json.load(open('$O/bench_c3_parts$parts.json')); print('parts $parts: %.4g steps/s' % d['value'], 'frac %.3f' % d['roofline']['frac'])" | tee -a $O/log.txt
done
bash tools/experiments/c3_trace.sh > $O/c3_trace.txt 2>&1; cat $O/c3_trace.txt | tee -a $O/log.txt
