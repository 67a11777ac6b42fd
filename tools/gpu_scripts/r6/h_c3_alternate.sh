#!/bin/bash
# round 6: config 3 (dense metric) with the two half-batches in strict alternation (one half multiplies while the other walks its trees)
O=gpurun_out/r6h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_configs.py tests/test_gpu_engines.py -m gpu -q -x -k "dense or config3" 2>&1 | tail -5 | tee $O/pytest_c3.log
for v in "alternate DHMC_NOTHING=1" "free_running DHMC_DENSE=alternate=0" "alternate_again DHMC_NOTHING=1"; do
  set -- $v
  r=$(env $2 timeout 600 python bench.py --config 3 --steps 3 --warmup 1 --transitions 100 --no-cpu-baseline 2>$O/err_$1.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3: %.4g leapfrog-steps/s, %.1f ms/step, frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']), d.get('rounds'))")
  echo "$1: $r" | tee -a $O/c3.txt
done
bash tools/experiments/c3_trace.sh > $O/trace_alternate.txt 2>&1; cp gpurun_out/c3trace/timeline.txt $O/timeline_alternate.txt; tail -12 $O/trace_alternate.txt
DHMC_DENSE=alternate=0 bash tools/experiments/c3_trace.sh > $O/trace_free.txt 2>&1; cp gpurun_out/c3trace/timeline.txt $O/timeline_free.txt; tail -6 $O/trace_free.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/c3prof -o c3 -- python $GRAFT_REPO_ROOT/bench.py --config 3 --steps 3 --warmup 1 --transitions 100 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_c3_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/err_c3prof.txt
cd $GRAFT_REPO_ROOT; f=$(find /tmp/c3prof -name '*kernel_stats.csv' | head -1); cp $f $O/c3_kernel_stats.csv; head -6 $O/c3_kernel_stats.csv | cut -c1-150
