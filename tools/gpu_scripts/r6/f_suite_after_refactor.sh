#!/bin/bash
# round 6: the whole GPU suite after dhmc_run was split into capi_run.hip (hybrid rounds, hipGraph option, leapfrog budget removed); config 5 with the
# anti-phase start of every second workgroup of a CU
O=gpurun_out/r6f; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -25 > $O/pytest.log; cat $O/pytest.log
for v in "library DHMC_NOTHING=1" "antiphase DHMC_LIB_PATH=tools/experiments/_v/antiphase/libdhmc_amd.so" "library_again DHMC_NOTHING=1"; do
  set -- $v
  r=$(env $2 timeout 600 python bench.py --config 5 --steps 5 --warmup 1 --no-cpu-baseline 2>$O/err_c5_$1.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5: %.4g leapfrog-steps/s, %.1f ms/step, frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))")
  echo "$1: $r" | tee -a $O/c5.txt
done
