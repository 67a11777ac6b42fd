#!/bin/bash
# round 6: config 3 with three launches before a round's product instead of six (the paired list product, K0 + list reset + K2 in one kernel)
O=gpurun_out/r6l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_configs.py tests/test_gpu_engines.py tests/test_gpu_user_functor.py -m gpu -q -x -k "dense or config3" 2>&1 | tail -5 | tee $O/pytest_c3.log
for v in "parts2 DHMC_NOTHING=1" "parts1 DHMC_DENSE=parts=1" "parts2_again DHMC_NOTHING=1"; do
  set -- $v
  r=$(env $2 timeout 600 python bench.py --config 3 --steps 3 --warmup 1 --transitions 100 --no-cpu-baseline 2>$O/err_$1.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3: %.4g leapfrog-steps/s, %.1f ms/step, frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))")
  echo "$1: $r" | tee -a $O/c3.txt
done
bash tools/experiments/c3_trace.sh > $O/trace.txt 2>&1; cp gpurun_out/c3trace/timeline.txt $O/timeline.txt; tail -6 $O/trace.txt
