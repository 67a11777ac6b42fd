#!/bin/bash
# round 5: where the headline kernel's L2 misses go — the latency of the L2's fabric requests against two calibration kernels
# (HBM stream, Infinity-Cache-resident copy)
O=$PWD/gpurun_out/r5dst; mkdir -p $O
REPO=$PWD
export TMPDIR=/tmp
cd /tmp
PMC="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum"
rm -rf /tmp/dst_cal /tmp/dst_bench /tmp/dst_c4
timeout 600 rocprofv3 --output-format csv --pmc $PMC -d /tmp/dst_cal -o cal -- python $REPO/tools/experiments/ea_latency_calibration.py > $O/cal.out 2> $O/cal.err
timeout 900 rocprofv3 --output-format csv --pmc $PMC -d /tmp/dst_bench -o bench -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --traffic none --transitions 100 > $O/bench.out 2> $O/bench.err
cd $REPO
for d in cal bench; do
  f=$(find /tmp/dst_$d -name "*counter_collection.csv" | head -1)
  echo "== $d ($f)" | tee -a $O/summary.txt
  python tools/summarize_ea_latency.py $f 2>&1 | tee -a $O/summary.txt
done
tail -2 $O/cal.err $O/bench.err
