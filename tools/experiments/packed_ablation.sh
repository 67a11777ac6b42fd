#!/bin/bash
# What each piece of the packed kernel's trip costs: builds of the funnel family with one piece cut out (timing only — the results of
# such a build are not the sampler's), run on the stuck-chain probe (every tree to the depth limit: all chains busy on every trip).
#   build (container):  bash tools/experiments/packed_ablation.sh build
#   run   (GPU box):    bash tools/experiments/packed_ablation.sh
set -e
ROOT=$(cd $(dirname $0)/../.. && pwd)
VARIANTS="BASE NO_RANDEXP NO_LOGADDEXP NO_MERGE_VEC NO_EXP NO_RANDEXP+NO_LOGADDEXP+NO_MERGE_VEC+NO_EXP"
if [ "${1:-run}" = build ]; then
    cd $ROOT/dynamichmc.jl_amd/csrc
    make -j8 >/dev/null
    mkdir -p $ROOT/tools/experiments/_abl
    FL="-O3 -std=c++17 -ffp-contract=off -fPIC --offload-arch=gfx950 -Wno-unused-result"
    OBJS=$(ls ../lib/obj/*.o | grep -v family_FunnelT)
    for v in $VARIANTS; do
        D=""; for f in $(echo $v | tr '+' ' '); do [ $f = BASE ] || D="$D -DPK_ABL_$f"; done
        ( /opt/rocm/bin/hipcc $FL $D -DDHMC_FAMILY=FunnelT -c -o $ROOT/tools/experiments/_abl/f_$v.o family.hip &&
          /opt/rocm/bin/hipcc $FL -shared -o $ROOT/tools/experiments/_abl/lib_$v.so $OBJS $ROOT/tools/experiments/_abl/f_$v.o -lhiprtc &&
          rm $ROOT/tools/experiments/_abl/f_$v.o ) &
    done
    wait
    ls -la $ROOT/tools/experiments/_abl
    exit 0
fi
for cpl in 2 4; do
  for v in $VARIANTS; do
    echo -n "cpl $cpl  $v: "
    DHMC_PACKED=1 DHMC_PK=cpl=$cpl PH_STUCK=1 DHMC_LIB_PATH=$ROOT/tools/experiments/_abl/lib_$v.so python $ROOT/tools/experiments/packed_probe.py ${2:-8} ${3:-10} 2>&1 | grep chains
  done
done
