#!/bin/bash
# round 6: v_mfma_f64_16x16x4_f64 against the number of independent accumulators per wave and waves per SIMD
O=gpurun_out/r6r; mkdir -p $O
hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_chain tools/experiments/mfma_f64_chain.hip && timeout 120 /tmp/mfma_chain | tee $O/mfma_chain.txt
