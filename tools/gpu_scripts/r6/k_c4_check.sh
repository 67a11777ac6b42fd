#!/bin/bash
# round 6: why does the default line's c4_32768 leg (1.08e9 at N = 1000) differ from the stand-alone run (1.51e9)?
O=gpurun_out/r6k; mkdir -p $O
for v in "default DHMC_NOTHING=1" "cpl4 DHMC_PK=cpl=4,max_waves=1024"; do
  set -- $v
  r=$(env $2 DHMC_DEBUG_ORDER=1 timeout -s KILL 300 python bench.py --config 4 --chains 32768 --steps 2 --warmup 1 --no-cpu-baseline --config-n 1000 2>$O/err_$1.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g (steps of 20)' % d['value'], '%.4g at N=1000, %.0f ms' % (d['at_config_n']['value'], d['at_config_n']['ms_per_step']))")
  echo "c4_32768 $1: $r" | tee -a $O/c4.txt
  grep "engine\|end game\|launch order" $O/err_$1.txt | tail -8
done
timeout -s KILL 900 python -m pytest tests/test_gpu_packed.py tests/test_gpu_user_functor.py -m gpu -x -q 2>&1 | tail -3 | tee $O/packed_tests.txt
