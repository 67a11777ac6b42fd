#!/bin/bash
set -u
O=$PWD/gpurun_out/r3i; mkdir -p $O
REPO=$PWD
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pf; timeout 500 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/pf -o t -- python $REPO/bench.py --steps 3 --warmup 1 --transitions 100 --no-cpu-baseline > $O/bench.json 2> $O/err.txt
f=$(find /tmp/pf -name '*kernel_stats.csv' | head -1); cp $f $O/kernel_stats.csv
head -8 $O/kernel_stats.csv | cut -c1-220
