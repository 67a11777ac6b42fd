#!/bin/bash
# round 6: the library's 64x64-tile product kernel with 8, 16, 32, 64 k per LDS stage (VGPR-form build)
O=gpurun_out/r6w; mkdir -p $O
cd tools/experiments; hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -o /tmp/gemm_vgpr gemm_v2_bench.hip 2>/dev/null; cd ../..
GEMM_BENCH_LIBRARY_TK=1 timeout 300 /tmp/gemm_vgpr | grep -v "v2 " | tee $O/gemm_tk.txt
