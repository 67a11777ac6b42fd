#!/bin/bash
# round 6: the library built with -mllvm -amdgpu-mfma-vgpr-form (matrix accumulators in architectural VGPRs): parity of the MFMA paths, configs 3 and 5
O=gpurun_out/r6s; mkdir -p $O
timeout -s KILL 1200 python -m pytest tests/test_gpu_dense.py tests/test_gpu_configs.py tests/test_gpu_engines.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt
for c in 3 5; do
  r=$(timeout -s KILL 400 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline 2>$O/err_c$c.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g leapfrog-steps/s, %.1f ms/step, frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))")
  echo "c$c: $r" | tee -a $O/bench.txt
done
export TMPDIR=/tmp; REPO=$PWD; cd /tmp
for c in 3 5; do
  rm -rf /tmp/pk$c; timeout -s KILL 300 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/pk$c -o t -- python $REPO/bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > $REPO/$O/bench_c${c}_under_rocprof.json 2> /tmp/pk$c.err
  f=$(find /tmp/pk$c -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $REPO/$O/c${c}_kernel_stats.csv
  head -5 $REPO/$O/c${c}_kernel_stats.csv | cut -c1-180
done
