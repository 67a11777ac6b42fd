#!/bin/bash
# round 6: the packed engine widened — the tridiagonal-precision normal, and a caller's functor (hiprtc) through the packed kernel
O=gpurun_out/r6j; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_user_functor.py tests/test_gpu_pipeline.py tests/test_gpu_sample_correctness.py -m gpu -q -x --durations=6 2>&1 | tail -25 | tee $O/pytest.log
