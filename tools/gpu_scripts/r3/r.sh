#!/bin/bash
export PYTHONPATH=tests
O=gpurun_out/r3r; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_gpu_dense.py tests/test_golden.py tests/test_gpu_engines.py tests/test_gpu_tolerance.py -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee $O/log.txt; tail -12 $O/pytest.txt | tee -a $O/log.txt
for f in 1 0; do
  DHMC_FUSE_K2=$f timeout -s KILL 300 python bench.py --config 3 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_c3_fuse$f.json
  python -c "
import json; d = json.load(open('$O/bench_c3_fuse$f.json')); print('fuse $f: %.4g steps/s' % d['value'], 'frac %.3f' % d['roofline']['frac'], d['tree'])" | tee -a $O/log.txt
done
