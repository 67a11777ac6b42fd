#!/bin/bash
export PYTHONPATH=tests
O=gpurun_out/r3t; mkdir -p $O
timeout -s KILL 300 python tools/fuzz_parity.py 150 11 > $O/fuzz.txt 2>&1; echo "fuzz rc=$?" | tee $O/log.txt; tail -4 $O/fuzz.txt | tee -a $O/log.txt
timeout -s KILL 2400 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee -a $O/log.txt; tail -6 $O/pytest.txt | tee -a $O/log.txt
