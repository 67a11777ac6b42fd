#!/bin/bash
# round 4: the two-waves-per-SIMD layout of the wide chain (trajectory rows in the workspace, 16 KB of LDS, 256 registers):
# parity of one golden + bench against the one-wave layout
O=$PWD/gpurun_out/r4p; mkdir -p $O
DHMC_L1_LDS=0 timeout -s KILL 200 python -m pytest tests/test_golden.py tests/test_gpu_engines.py -m gpu -q -x 2>&1 | tail -3
for v in 1 0 1 0; do
  DHMC_L1_LDS=$v timeout -s KILL 200 python bench.py --steps 5 --warmup 2 --transitions 200 --no-cpu-baseline --no-other-configs --traffic none 2>/dev/null | tail -1 | python -c "
import json,sys; d = json.loads(sys.stdin.read()); print('L1_LDS=$v %.4g' % d['value'], 'ms/step %.2f' % d['ms_per_step'])"
done
