#!/bin/bash
# round 6, second session: the logistic round's two memsets and its list kernel folded into K1 / K2: parity, same-box A/B against the build before
O=gpurun_out/r6bt; mkdir -p $O
timeout -s KILL 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_engines.py tests/test_gpu_external.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt
bash tools/gpu_scripts/r6b/n_gemm_ab.sh prev "5"
