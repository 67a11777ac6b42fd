#!/bin/bash
export PYTHONPATH=tests
O=gpurun_out/r3u; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee $O/log.txt; tail -12 $O/pytest.txt | tee -a $O/log.txt
