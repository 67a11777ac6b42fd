#!/bin/bash
# round 6: from how many chains on does the packed kernel beat the wave kernel on the dense-precision normal?
O=gpurun_out/r6p; mkdir -p $O
timeout -s KILL 300 python -m pytest tests/test_gpu_packed.py -m gpu -x -q -k "normal_families" 2>&1 | tail -2 | tee $O/pytest.txt
for a in "32 2048 200" "32 4096 200" "32 8192 200" "16 2048 200" "16 4096 200" "8 2048 200" "8 4096 200" "24 4096 200"; do
  timeout -s KILL 200 python tools/experiments/packed_dense_normal_probe.py $a 2>/dev/null | tee -a $O/probe.txt
done
