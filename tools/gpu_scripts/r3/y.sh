#!/bin/bash
# round 3: user functor with the dense metric (run-time compiled round engine), config-3 kernel stats of the final fused build
O=gpurun_out/r3y; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_user_functor.py -x -q 2>&1 | tail -8 > $O/pytest_user.txt
REPO=$PWD
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/c3s
timeout 500 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/c3s -o t -- python $REPO/bench.py --config 3 --steps 3 --warmup 1 > $REPO/$O/c3_under_rocprof.json 2> /dev/null
cp $(find /tmp/c3s -name '*kernel_stats.csv' | head -1) $REPO/$O/c3_kernel_stats.csv
cd $REPO
python bench.py --config 3 --steps 5 --warmup 1 2>/dev/null | tail -1 > $O/bench_c3.json
cat $O/pytest_user.txt; head -12 $O/c3_kernel_stats.csv | cut -c1-200; cat $O/bench_c3.json | cut -c1-300
