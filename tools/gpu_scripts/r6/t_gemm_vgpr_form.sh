#!/bin/bash
# round 6: the fp64 GEMM kernels (library <2,2,16> and the v2 tile shapes) compiled with and without -mllvm -amdgpu-mfma-vgpr-form
O=gpurun_out/r6t; mkdir -p $O
cd tools/experiments
hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o /tmp/gemm_agpr gemm_v2_bench.hip 2>/dev/null
hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -o /tmp/gemm_vgpr gemm_v2_bench.hip 2>/dev/null
cd ../..
echo "== compiler's choice (AGPR accumulators)" | tee $O/gemm.txt; timeout 300 /tmp/gemm_agpr | tee -a $O/gemm.txt
echo "== -mllvm -amdgpu-mfma-vgpr-form" | tee -a $O/gemm.txt; timeout 300 /tmp/gemm_vgpr | tee -a $O/gemm.txt
