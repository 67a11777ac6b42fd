#!/bin/bash
# round 6, second session: fixed cost against steady state of gemm_rows_f64_kernel<2,2,16>, and ablations of its k-loop
O=gpurun_out/r6bb; mkdir -p $O
cd tools/experiments
hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -o /tmp/gemm_ramp gemm_ramp_probe.hip 2>/dev/null
cd ../..
timeout 300 /tmp/gemm_ramp | tee $O/ramp.txt
