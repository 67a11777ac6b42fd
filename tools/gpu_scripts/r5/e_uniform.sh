#!/bin/bash
O=gpurun_out/r5e; mkdir -p $O
(for sc in "30 4096 200" "30 32768 200" "8 32768 200" "60 16384 100" "16 65536 100"; do timeout 300 python tools/experiments/packed_uniform_probe.py $sc; done) 2>&1 | grep -v amdgpu.ids > $O/uniform.txt
cat $O/uniform.txt
