#!/bin/bash
# round 6: config 5 with the fused kernel's products as 4x4x4 matrix instructions (eight chains per wave); then the whole GPU suite
O=gpurun_out/r6g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_engines.py tests/test_gpu_tolerance.py tests/test_gpu_external.py -m gpu -q -x -k "logistic or config5" 2>&1 | tail -5 | tee $O/pytest_c5.log
timeout 600 python bench.py --config 5 --steps 5 --warmup 1 --no-cpu-baseline --config-n 200 2>$O/err_c5.txt | tail -1 > $O/bench_c5.json; python -c "
import json; d=json.load(open('$O/bench_c5.json')); print('c5: %.4g leapfrog-steps/s, %.1f ms/step, frac %.3f; at N=200 %.4g' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['at_config_n']['value']))"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/c5prof -o c5 -- python $GRAFT_REPO_ROOT/bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_c5_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/err_c5prof.txt
cd $GRAFT_REPO_ROOT; f=$(find /tmp/c5prof -name '*kernel_stats.csv' | head -1); cp $f $O/c5_kernel_stats.csv; head -5 $O/c5_kernel_stats.csv | cut -c1-150
timeout 1700 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -25 > $O/pytest.log; cat $O/pytest.log
