#!/bin/bash
export PYTHONPATH=tests
O=gpurun_out/r3k; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_gpu_engines.py tests/test_golden.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee $O/log.txt; tail -4 $O/pytest.txt | tee -a $O/log.txt
python tools/experiments/tpl_tridiag_time.py 2>/dev/null | tee -a $O/log.txt
