#!/bin/bash
# Like build_variant_fast.sh for the ABI's own translation unit: recompiles dhmc_capi.hip and capi_run.hip (dhmc_run and its engines) with extra flags and links it with the
# library's other objects.   usage: bash tools/experiments/build_variant_capi.sh <name> [flags]
# e.g.  … gemmv2_3 -DDHMC_GEMM_ROWS_V2=3 -include $PWD/tools/experiments/gemm_f64_v2.hpp
set -e
ROOT=$(cd $(dirname $0)/../.. && pwd)
NAME=$1; shift
D=$ROOT/tools/experiments/_v/$NAME
mkdir -p $D
OBJ=$ROOT/dynamichmc.jl_amd/lib/obj
FLAGS="-O3 -std=c++17 -ffp-contract=off -fPIC --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -Wno-unused-result"
(cd $ROOT/dynamichmc.jl_amd/csrc && /opt/rocm/bin/hipcc $FLAGS "$@" -c -o $D/dhmc_capi.o dhmc_capi.hip && /opt/rocm/bin/hipcc $FLAGS "$@" -c -o $D/capi_run.o capi_run.hip)
OTHERS=$(ls $OBJ/*.o | grep -v "dhmc_capi.o\|capi_run.o")
/opt/rocm/bin/hipcc $FLAGS -shared -o $D/libdhmc_amd.so $OTHERS $D/dhmc_capi.o $D/capi_run.o -lhiprtc
rm -f $D/dhmc_capi.o $D/capi_run.o
ls -la $D/libdhmc_amd.so
