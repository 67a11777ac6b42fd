#!/bin/bash
# round 3, GPU call B: the one-product dense recurrence — parity (dense / golden / engines / tolerance), config 3 bench in both modes.
set -u
O=gpurun_out/r3b; mkdir -p $O
export PYTHONPATH=tests
timeout -s KILL 900 python -m pytest tests/test_golden.py tests/test_gpu_dense.py tests/test_gpu_engines.py tests/test_gpu_tolerance.py -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee $O/log.txt
tail -25 $O/pytest.txt | tee -a $O/log.txt
for p in 1 2; do
  DHMC_DENSE_PRODUCTS=$p timeout -s KILL 300 python bench.py --config 3 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_c3_p$p.json
  python -c "
import json; d = json.load(open('$O/bench_c3_p$p.json')); print('products $p: %.4g steps/s' % d['value'], 'frac %.3f' % d['roofline']['frac'], d['tree'])" | tee -a $O/log.txt
done
