#!/bin/bash
# round 6, second session: the product kernel beside a kernel that saturates HBM, 1-4 k-tiles of loads in flight
O=gpurun_out/r6bf; mkdir -p $O
cd tools/experiments
hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -o /tmp/gemm_cont gemm_contention_probe.hip 2>/dev/null
cd ../..
timeout 300 /tmp/gemm_cont | tee $O/contention.txt
