#!/bin/bash
# round 5: gate width (DHMC_PK_ALIGN: transitions start on trips = 0 mod A, which aligns the chains' merge cascades below level log2 A)
# and coordinates per lane on the throughput-bound launch: 32768 funnel chains through the packed kernel's queue
O=gpurun_out/r5x; mkdir -p $O
run() {  # chains name env
  env ${3//,/ } DHMC_HYBRID_MIN_CHAINS=8192 DHMC_DEBUG_ORDER=1 timeout 600 python bench.py --config 4 --chains $1 --transitions 1000 --steps 1 --warmup 0 2> $O/c4_$1_$2.err | tail -1 > $O/c4_$1_$2.json
  python -c "
import json; d = json.load(open('$O/c4_$1_$2.json')); print('$1 $2: %.4g' % d['value'], 'ms %.0f' % d['ms_per_step'])"
  grep "round" $O/c4_$1_$2.err | tail -40 | cut -c1-200 > $O/c4_$1_$2.rounds; tail -${4:-2} $O/c4_$1_$2.rounds
}
for a in 1 2 4 8 16 32 64; do run 32768 packed_a$a DHMC_PACKED=1,DHMC_PK_ALIGN=$a 0; done
for a in 4 16; do run 32768 packed_cpl2_a$a DHMC_PACKED=1,DHMC_PK_ALIGN=$a,DHMC_PK_CPL=2 0; done
for a in 8 16 32; do run 32768 r8_b8_a$a DHMC_HYBRID_BUDGET=8,DHMC_PK_ALIGN=$a; done
run 32768 r8_b8_a16_cpl2 DHMC_HYBRID_BUDGET=8,DHMC_PK_ALIGN=16,DHMC_PK_CPL=2
