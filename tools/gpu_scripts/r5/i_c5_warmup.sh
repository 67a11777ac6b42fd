#!/bin/bash
O=gpurun_out/r5i; mkdir -p $O
timeout 900 python tools/experiments/c5_warmup_probe.py 2>&1 | grep -v amdgpu.ids > $O/c5_warmup.txt; cat $O/c5_warmup.txt
