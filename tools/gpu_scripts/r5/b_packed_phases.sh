#!/bin/bash
# round 5: where the packed kernel's clocks go (instrumented build), config 4's chains
O=gpurun_out/r5b; mkdir -p $O
export DHMC_LIB_PATH=$PWD/tools/experiments/_phase/libdhmc_amd_FunnelT.so
(for a in 4 1 16; do echo "== align $a"; DHMC_PK_ALIGN=$a timeout 300 python tools/experiments/packed_phase_timing.py 4096 100; done
 echo "== 8 chains (one wave)"; timeout 300 python tools/experiments/packed_phase_timing.py 8 100
 echo "== wave-per-chain kernel"; DHMC_PACKED=0 PH_D=30 PH_TARGET=funnel timeout 300 python tools/experiments/phase_timing.py 4096 100) > $O/phases.txt 2>&1
cat $O/phases.txt
