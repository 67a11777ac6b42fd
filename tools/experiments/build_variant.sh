#!/bin/bash
# Builds the library from the working tree into tools/experiments/_v/<name>/libdhmc_amd.so (own object directory), so that
# several versions of the kernels can be timed in one GPU call: DHMC_LIB_PATH=tools/experiments/_v/<name>/libdhmc_amd.so.
# usage: bash tools/experiments/build_variant.sh <name> [extra hipcc flags]
set -e
ROOT=$(cd $(dirname $0)/../.. && pwd)
NAME=$1; shift
D=$ROOT/tools/experiments/_v/$NAME
mkdir -p $D/obj
make -j8 -C $ROOT/dynamichmc.jl_amd/csrc OUT=$D/libdhmc_amd.so OBJDIR=$D/obj \
    HIPFLAGS="-O3 -std=c++17 -ffp-contract=off -fPIC --offload-arch=gfx950 -Wno-unused-result $*" 2>&1 | grep -i 'error' || true
rm -rf $D/obj
ls -la $D/libdhmc_amd.so
