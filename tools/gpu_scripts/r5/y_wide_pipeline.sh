#!/bin/bash
# round 5: the pipeline kernel for rows of 128 / 256 doubles — parity, then the latency of a handful of chains (the reference's use)
O=gpurun_out/r5p2; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_pipeline.py -x -q 2>&1 | tail -12 > $O/tests.log; cat $O/tests.log
for d in 100 256; do
  for v in "wave DHMC_PIPELINE=0,DHMC_PACKED=0" "pipeline DHMC_PIPELINE=1"; do
    set -- $v
    env ${2//,/ } timeout 90 python tools/experiments/few_chain_latency.py $d 4 $1 2>&1 | grep chains | tee -a $O/latency.txt
  done
done
env PH_FUNNEL=1 DHMC_PIPELINE=0 DHMC_PACKED=0 timeout 90 python tools/experiments/few_chain_latency.py 100 4 wave_fun 2>&1 | grep chains | tee -a $O/latency.txt
env PH_FUNNEL=1 DHMC_PIPELINE=1 timeout 90 python tools/experiments/few_chain_latency.py 100 4 pipe_fun 2>&1 | grep chains | tee -a $O/latency.txt
