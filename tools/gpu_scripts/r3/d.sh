#!/bin/bash
set -u
O=gpurun_out/r3d; mkdir -p $O
export PYTHONPATH=tests
timeout -s KILL 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dense.py tests/test_gpu_cabi.py tests/test_gpu_tolerance.py -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee $O/log.txt
tail -15 $O/pytest.txt | tee -a $O/log.txt
for T in 20 100; do timeout -s KILL 300 python tools/pcie_rate.py $T 2>/dev/null | tail -1 | tee -a $O/pcie_rate.txt; done
timeout -s KILL 300 python bench.py --config 3 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_c3.json
python -c "
import json; d = json.load(open('$O/bench_c3.json')); print('config 3: %.4g steps/s' % d['value'], 'frac %.3f' % d['roofline']['frac'])" | tee -a $O/log.txt
