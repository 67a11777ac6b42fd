#!/bin/bash
# Region-by-region clock breakdown of nuts_run_kernel<StdNormalT,16,true> (BASELINE config 2's kernel).
# Builds tools/experiments/_phase/libdhmc_amd.so = the regular objects with the StdNormalT family recompiled
# with -DDHMC_PHASE_TIMING (run on the build container: `bash tools/experiments/phase_timing.sh build`), then on the GPU
# box: `bash tools/experiments/phase_timing.sh`.
set -e
ROOT=$(cd $(dirname $0)/../.. && pwd)
FAM=${FAM:-StdNormalT}     # FAM=FunnelT: the funnel family instrumented instead (PH_D=30 PH_TARGET=funnel for phase_timing.py)
if [ "${1:-run}" = build ]; then
    cd $ROOT/dynamichmc.jl_amd/csrc
    make -j8 >/dev/null
    mkdir -p $ROOT/tools/experiments/_phase
    OUTSO=$ROOT/tools/experiments/_phase/libdhmc_amd${FAM:+_$FAM}.so
    FL="-O3 -std=c++17 -ffp-contract=off -fPIC --offload-arch=gfx950 -Wno-unused-result"
    /opt/rocm/bin/hipcc $FL -DDHMC_PHASE_TIMING -DDHMC_FAMILY=$FAM -c -o $ROOT/tools/experiments/_phase/family_$FAM.o family.hip
    OBJS=$(ls ../lib/obj/*.o | grep -v family_$FAM)
    /opt/rocm/bin/hipcc $FL -shared -o $OUTSO $OBJS $ROOT/tools/experiments/_phase/family_$FAM.o -lhiprtc
    rm $ROOT/tools/experiments/_phase/family_$FAM.o
    exit 0
fi
DHMC_LIB_PATH=$ROOT/tools/experiments/_phase/libdhmc_amd_$FAM.so python $ROOT/tools/experiments/phase_timing.py "$@"
