#!/bin/bash
# round 5: what the engines deliver on 32768 funnel chains when no chain can hold the launch open
O=gpurun_out/r5y; mkdir -p $O
for v in "wave DHMC_PACKED=0,DHMC_PIPELINE=0" "packed_cpl4_a4 DHMC_PACKED=1" "packed_cpl4_a16 DHMC_PACKED=1,DHMC_PK_ALIGN=16" "packed_cpl4_a1 DHMC_PACKED=1,DHMC_PK_ALIGN=1" \
         "packed_cpl2_a4 DHMC_PACKED=1,DHMC_PK_CPL=2" "packed_cpl4_noqueue DHMC_PACKED=1,DHMC_PK_QUEUE=0" "packed_cpl4_w512 DHMC_PACKED=1,DHMC_PK_MAX_WAVES=512" \
         "packed_cpl4_lds0 DHMC_PACKED=1,DHMC_PK_LDS_LEVELS=0"; do
  set -- $v
  env ${2//,/ } DHMC_HYBRID=0 timeout 300 python tools/experiments/packed_bulk_probe.py 32768 $1 2>&1 | grep chains | tee -a $O/bulk.txt
done
env DHMC_PACKED=1 PH_DEPTH=8 DHMC_HYBRID=0 timeout 300 python tools/experiments/packed_bulk_probe.py 32768 packed_cpl4_depth8 2>&1 | grep "depth" | tee -a $O/bulk.txt
env DHMC_PACKED=0 DHMC_PIPELINE=0 PH_DEPTH=8 DHMC_HYBRID=0 timeout 300 python tools/experiments/packed_bulk_probe.py 32768 wave_depth8 2>&1 | grep "depth" | tee -a $O/bulk.txt
