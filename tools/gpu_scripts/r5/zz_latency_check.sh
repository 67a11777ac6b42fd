#!/bin/bash
# round 5: latency per leapfrog of the pipeline kernel on the final build (stuck chains: 8 chains x 10 transitions x 1023 leapfrogs)
for v in "pipeline DHMC_PIPELINE=1" "wave DHMC_PIPELINE=0,DHMC_PACKED=0"; do
  set -- $v
  for r in 1 2; do env ${2//,/ } PH_STUCK=1 timeout 100 python tools/experiments/packed_probe.py 8 10 2>&1 | grep chains | sed "s/^/$1 /"; done
done
