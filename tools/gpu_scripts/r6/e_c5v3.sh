#!/bin/bash
# round 6: config 5, the fused eta + link kernel with the link's pieces between the matrix instructions; the end game as shipped; packed occupancy 2 by default
O=gpurun_out/r6e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_engines.py tests/test_gpu_tolerance.py tests/test_gpu_external.py -m gpu -q -x -k "logistic or config5" 2>&1 | tail -5 | tee $O/pytest_c5.log
timeout 600 python bench.py --config 5 --steps 5 --warmup 1 --no-cpu-baseline 2>$O/err_c5.txt | tail -1 > $O/bench_c5.json; python -c "
import json; d=json.load(open('$O/bench_c5.json')); print('c5: %.4g leapfrog-steps/s, %.1f ms/step, frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/c5prof -o c5 -- python $GRAFT_REPO_ROOT/bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/$O/err_c5prof.txt
cd $GRAFT_REPO_ROOT; f=$(find /tmp/c5prof -name '*kernel_stats.csv' | head -1); cp $f $O/c5_kernel_stats.csv; head -4 $O/c5_kernel_stats.csv | cut -c1-150
timeout 900 python -m pytest tests/test_gpu_packed.py -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest_packed.log
