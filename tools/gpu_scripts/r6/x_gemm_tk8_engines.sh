#!/bin/bash
# round 6: configs 3 and 5 with 8 instead of 16 k per LDS stage in the product kernel (variant libraries: tools/experiments/_v/tk8, tk16)
O=gpurun_out/r6x; mkdir -p $O
for v in tk16 tk8 tk16 tk8; do
  for c in 3 5; do
    r=$(DHMC_LIB_PATH=$PWD/tools/experiments/_v/$v/libdhmc_amd.so timeout -s KILL 400 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline 2>$O/err_${v}_c$c.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g leapfrog-steps/s, %.1f ms/step' % (d['value'], d['ms_per_step']))")
    echo "$v c$c: $r" | tee -a $O/bench.txt
  done
done
DHMC_LIB_PATH=$PWD/tools/experiments/_v/tk8/libdhmc_amd.so timeout -s KILL 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -2 | tee $O/pytest_tk8.txt
