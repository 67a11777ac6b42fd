#!/bin/bash
# round 6, second session: the logistic link with both quotients from one reciprocal and the special cases behind one ballot —
# parity (scalar-math self-test kinds 10 / 11, config 5's bit-exact tests, the logistic engine tests), config 5's bench and kernel stats
O=gpurun_out/r6ba; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_gpu_detmath.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_detmath.txt
timeout -s KILL 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_engines.py -m gpu -x -q -k "logistic or config5 or c5" 2>&1 | tail -3 | tee $O/pytest_c5.txt
for i in 1 2; do
  r=$(timeout -s KILL 400 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline 2>$O/err_c5.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g leapfrog-steps/s, %.1f ms/step, frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))")
  echo "c5: $r" | tee -a $O/bench.txt
done
export TMPDIR=/tmp; REPO=$PWD; cd /tmp
rm -rf /tmp/pk5; timeout -s KILL 300 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/pk5 -o t -- python $REPO/bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline > $REPO/$O/bench_c5_under_rocprof.json 2> /tmp/pk5.err
f=$(find /tmp/pk5 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $REPO/$O/c5_kernel_stats.csv
head -6 $REPO/$O/c5_kernel_stats.csv | cut -c1-200
