#!/bin/bash
# round 6, second session: gemm_rows_f64_kernel with two k-tiles of loads in flight: parity of the MFMA paths, configs 3 and 5
O=gpurun_out/r6bh; mkdir -p $O
timeout -s KILL 1500 python -m pytest tests/test_gpu_dense.py tests/test_gpu_configs.py tests/test_gpu_engines.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt
bash tools/gpu_scripts/r6b/g_c3_quick.sh r6bh "3 5"
