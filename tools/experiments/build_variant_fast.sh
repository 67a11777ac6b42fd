#!/bin/bash
# Like build_variant.sh, but recompiles ONE family object (default StdNormalT; FAM=… in the environment) with the extra
# flags and links it with the library's other, already built objects (dynamichmc.jl_amd/lib/obj — run `make` first):
# ≈ 40 s instead of 3 min per variant.   usage: [FAM=TridiagNormalT] bash tools/experiments/build_variant_fast.sh <name> [flags]
set -e
ROOT=$(cd $(dirname $0)/../.. && pwd)
NAME=$1; shift
FAM=${FAM:-StdNormalT}
D=$ROOT/tools/experiments/_v/$NAME
mkdir -p $D
OBJ=$ROOT/dynamichmc.jl_amd/lib/obj
FLAGS="-O3 -std=c++17 -ffp-contract=off -fPIC --offload-arch=gfx950 -Wno-unused-result"
(cd $ROOT/dynamichmc.jl_amd/csrc && /opt/rocm/bin/hipcc $FLAGS -DDHMC_FAMILY=$FAM "$@" -c -o $D/family_$FAM.o family.hip)
OTHERS=$(ls $OBJ/*.o | grep -v "family_$FAM.o")
/opt/rocm/bin/hipcc $FLAGS -shared -o $D/libdhmc_amd.so $OTHERS $D/family_$FAM.o -lhiprtc
rm -f $D/family_$FAM.o
ls -la $D/libdhmc_amd.so
