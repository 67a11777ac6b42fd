#!/bin/bash
# round 6: config 5's fused eta + link kernel with the roles split over the workgroup's waves (DHMC_LOGISTIC_ROLES=1) against the one-role kernel
O=gpurun_out/r6q; mkdir -p $O
DHMC_LOGISTIC_ROLES=1 timeout -s KILL 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_engines.py -m gpu -x -q -k "logistic or config5" 2>&1 | tail -3 | tee $O/pytest_roles.txt
for v in "one_role DHMC_LOGISTIC_ROLES=0" "roles DHMC_LOGISTIC_ROLES=1" "one_role_again DHMC_LOGISTIC_ROLES=0" "roles_again DHMC_LOGISTIC_ROLES=1"; do
  set -- $v
  r=$(env $2 timeout -s KILL 400 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline 2>$O/err_$1.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5: %.4g leapfrog-steps/s, %.1f ms/step, frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))")
  echo "$1: $r" | tee -a $O/c5.txt
done
export TMPDIR=/tmp; REPO=$PWD; cd /tmp
rm -rf /tmp/pk5; DHMC_LOGISTIC_ROLES=1 timeout -s KILL 300 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/pk5 -o t -- python $REPO/bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline > $REPO/$O/bench_c5_roles_under_rocprof.json 2> /tmp/pk5.err
f=$(find /tmp/pk5 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $REPO/$O/c5_roles_kernel_stats.csv
head -6 $REPO/$O/c5_roles_kernel_stats.csv | cut -c1-200
