#!/bin/bash
# round 6, second session: the blocks' butterflies folded into K2 (no logistic_block_sums_kernel in the round engine): parity, same-box A/B
O=gpurun_out/r6bo; mkdir -p $O
timeout -s KILL 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_engines.py tests/test_gpu_external.py -m gpu -x -q -k "logistic or config5" 2>&1 | tail -3 | tee $O/pytest_c5.txt
bash tools/gpu_scripts/r6b/n_gemm_ab.sh gemm_late "5"
