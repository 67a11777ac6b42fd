#!/bin/bash
set -u
O=gpurun_out/r3e; mkdir -p $O
export PYTHONPATH=tests
timeout -s KILL 2400 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee $O/log.txt
tail -15 $O/pytest.txt | tee -a $O/log.txt
