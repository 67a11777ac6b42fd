#!/bin/bash
# round 6: the link's log part as selects instead of per-argument branches (config 5)
O=gpurun_out/r6y; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_engines.py tests/test_gpu_external.py -m gpu -x -q -k "logistic or config5" 2>&1 | tail -2 | tee $O/pytest.txt
for i in 1 2; do
  r=$(timeout -s KILL 400 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline 2>$O/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5: %.4g leapfrog-steps/s, %.1f ms/step, frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))")
  echo "$r" | tee -a $O/c5.txt
done
export TMPDIR=/tmp; REPO=$PWD; cd /tmp
rm -rf /tmp/pk5; timeout -s KILL 300 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/pk5 -o t -- python $REPO/bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline > $REPO/$O/bench_c5_under_rocprof.json 2> /tmp/pk5.err
f=$(find /tmp/pk5 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $REPO/$O/c5_kernel_stats.csv
head -4 $REPO/$O/c5_kernel_stats.csv | cut -c1-170
