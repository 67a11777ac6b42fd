#!/bin/bash
# round 5: where the headline kernel's L2 misses go — the latency of the L2's fabric requests against two calibration kernels
# (HBM stream, Infinity-Cache-resident copy).  Two counters per pass (five in one pass: "exceeds the capabilities of the hardware",
# and that rocprofv3 then hangs until killed); every pass under a hard timeout.
O=$PWD/gpurun_out/r5dst; mkdir -p $O; rm -f $O/summary.txt
REPO=$PWD
export TMPDIR=/tmp
cd /tmp
i=0
for PMC in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  rm -rf /tmp/dst_cal$i /tmp/dst_bench$i
  timeout -s KILL 100 rocprofv3 --output-format csv --pmc $PMC -d /tmp/dst_cal$i -o cal -- python $REPO/tools/experiments/ea_latency_calibration.py > $O/cal$i.out 2> $O/cal$i.err
  timeout -s KILL 150 rocprofv3 --output-format csv --pmc $PMC -d /tmp/dst_bench$i -o bench -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --traffic none --transitions 100 > $O/bench$i.out 2> $O/bench$i.err
done
cd $REPO
for d in cal bench; do
  echo "== $d" | tee -a $O/summary.txt
  python tools/summarize_ea_latency.py $(find /tmp/dst_${d}1 /tmp/dst_${d}2 /tmp/dst_${d}3 -name "*counter_collection.csv" | sort) 2>&1 | tee -a $O/summary.txt
done
grep -h "exceeds\|error" $O/*.err | head -5
