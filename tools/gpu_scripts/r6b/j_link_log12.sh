#!/bin/bash
# round 6, second session: the link's logarithm through the derived [1, 2) table — parity and config 5
O=gpurun_out/r6bj; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_gpu_detmath.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_detmath.txt
timeout -s KILL 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_engines.py -m gpu -x -q -k "logistic or config5" 2>&1 | tail -3 | tee $O/pytest_c5.txt
bash tools/gpu_scripts/r6b/g_c3_quick.sh r6bj "5"
