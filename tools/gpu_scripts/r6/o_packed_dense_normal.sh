#!/bin/bash
# round 6: the dense-precision normal in the packed kernel — parity (packed + user-functor + engines GPU tests) and throughput against the wave kernel
O=gpurun_out/r6o; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_gpu_packed.py tests/test_gpu_pipeline.py tests/test_gpu_engines.py -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.txt
for a in "8 16384 200" "16 16384 200" "32 16384 200" "64 8192 200" "32 1024 200"; do
  timeout -s KILL 200 python tools/experiments/packed_dense_normal_probe.py $a 2>/dev/null | tee -a $O/probe.txt
done
