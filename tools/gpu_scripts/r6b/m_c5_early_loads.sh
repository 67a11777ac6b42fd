#!/bin/bash
# round 6, second session: the fused kernel (next-stage loads before the barrier; then 128-k stages staged as two halves) — parity, config 5, the clocks
O=gpurun_out/r6bm; mkdir -p $O
timeout -s KILL 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_engines.py -m gpu -x -q -k "logistic or config5" 2>&1 | tail -3 | tee $O/pytest_c5.txt
bash tools/gpu_scripts/r6b/g_c3_quick.sh r6bm "5"
bash tools/gpu_scripts/r6b/l_c5_clocks.sh > /dev/null 2>&1; head -12 gpurun_out/r6bl/clocks.txt | cut -c1-160
