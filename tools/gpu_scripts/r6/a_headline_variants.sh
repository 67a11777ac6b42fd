#!/bin/bash
# round 6: the permuted-wave reduction (wave.hpp xl_reduce) — unit test against wave_allreduce, the headline kernel with / without it and with the
# trajectory's rho parked in accumulation registers, then the GPU suite with the library as built
O=gpurun_out/r6a; mkdir -p $O
tools/experiments/_v/xl_test 2>&1 | tee $O/xl_test.txt
for v in "library DHMC_NOTHING=1" "noxl DHMC_LIB_PATH=tools/experiments/_v/noxl/libdhmc_amd.so" "xltra DHMC_LIB_PATH=tools/experiments/_v/xltra/libdhmc_amd.so" "tra DHMC_LIB_PATH=tools/experiments/_v/tra/libdhmc_amd.so" "library_again DHMC_NOTHING=1" "noxl_again DHMC_LIB_PATH=tools/experiments/_v/noxl/libdhmc_amd.so"; do
  set -- $v
  r=$(env $2 timeout -s KILL 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other-configs --traffic none 2>$O/err_$1.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g leapfrog-steps/s' % d['value'], 'ms %.2f' % d['ms_per_step'], 'frac %.4f' % d['roofline']['frac'], d['roofline']['kernel'], d.get('tree'))")
  echo "$1: $r" | tee -a $O/variants.txt
done
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -30 > $O/pytest.log; cat $O/pytest.log
