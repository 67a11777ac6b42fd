#!/bin/bash
O=gpurun_out/r5m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_external.py tests/test_gpu_sample_correctness.py -q -k "symmetric or dense_metric_for_external or random_correlated or literal" 2>&1 | tail -8 > $O/log.txt; cat $O/log.txt
