#!/bin/bash
export PYTHONPATH=tests
O=gpurun_out/r3j; mkdir -p $O
python tools/experiments/tpl_tridiag_time.py 2>/dev/null | tee $O/log.txt
DHMC_LIB_PATH=$PWD/tools/experiments/_v/all_tpl/libdhmc_amd.so python tools/experiments/tpl_tridiag_time.py 2>/dev/null | tee -a $O/log.txt
