#!/bin/bash
export PYTHONPATH=tests
O=gpurun_out/r3x; mkdir -p $O
timeout -s KILL 600 python -m pytest tests/test_gpu_configs.py -m gpu -q -k "rccl or launches" > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee $O/log.txt; tail -15 $O/pytest.txt | tee -a $O/log.txt
