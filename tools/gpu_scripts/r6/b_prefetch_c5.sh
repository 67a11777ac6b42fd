#!/bin/bash
# round 6: (1) software prefetch variants of the headline kernel (workspace rows into L2, softplus rows into the scalar cache);
# (2) config 5 with the link fused into the eta GEMM: parity tests, bench line, kernel stats
O=gpurun_out/r6b; mkdir -p $O
for v in "base6 DHMC_LIB_PATH=tools/experiments/_v/base6/libdhmc_amd.so" "pf DHMC_LIB_PATH=tools/experiments/_v/pf/libdhmc_amd.so" "pfr DHMC_LIB_PATH=tools/experiments/_v/pfr/libdhmc_amd.so" "pfl DHMC_LIB_PATH=tools/experiments/_v/pfl/libdhmc_amd.so" "pftra DHMC_LIB_PATH=tools/experiments/_v/pftra/libdhmc_amd.so" "base6_again DHMC_LIB_PATH=tools/experiments/_v/base6/libdhmc_amd.so"; do
  set -- $v
  r=$(env $2 timeout -s KILL 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other-configs --traffic none 2>$O/err_$1.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g leapfrog-steps/s' % d['value'], 'ms %.2f' % d['ms_per_step'], 'frac %.4f' % d['roofline']['frac'])")
  echo "$1: $r" | tee -a $O/variants.txt
done
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_engines.py tests/test_gpu_tolerance.py tests/test_gpu_external.py -m gpu -q -x -k "logistic or config5" 2>&1 | tail -15 | tee $O/pytest_c5.log
timeout 600 python bench.py --config 5 --steps 5 --warmup 1 --no-cpu-baseline --config-n 200 2>$O/err_c5.txt | tail -1 > $O/bench_c5.json; cat $O/bench_c5.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/c5prof -o c5 -- python $GRAFT_REPO_ROOT/bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/$O/err_c5prof.txt
cd $GRAFT_REPO_ROOT; f=$(find /tmp/c5prof -name '*kernel_stats.csv' | head -1); cp $f $O/c5_kernel_stats.csv; head -12 $O/c5_kernel_stats.csv | cut -c1-160
