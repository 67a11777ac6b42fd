#!/bin/bash
O=$PWD/gpurun_out/r4o; mkdir -p $O
timeout -s KILL 300 python -m pytest tests/test_gpu_dense.py -m gpu -q -x 2>&1 | tail -3
for v in cur nopipe cur nopipe; do
  DHMC_LIB_PATH=$PWD/tools/experiments/_v/$v/libdhmc_amd.so timeout -s KILL 200 python bench.py --steps 5 --warmup 2 --transitions 200 --no-cpu-baseline --no-other-configs --traffic none 2>/dev/null | tail -1 | python -c "
import json,sys; d = json.loads(sys.stdin.read()); print('$v %.4g' % d['value'], 'ms/step %.2f' % d['ms_per_step'])"
done
