#!/bin/bash
# round 6: do fp64 matrix instructions and another wave's fp64 vector instructions overlap on a SIMD?
O=gpurun_out/r6v; mkdir -p $O
hipcc -O3 --offload-arch=gfx950 -o /tmp/mix tools/experiments/mfma_valu_f64_overlap.hip && timeout 120 /tmp/mix | tee $O/overlap.txt
