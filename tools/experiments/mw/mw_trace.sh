#!/bin/bash
# runs tools/experiments/mw_trace.py with the -DMW_TRACE build of the library (tools/experiments/_mwtrace/libdhmc_amd.so:
#   hipcc $HIPFLAGS -DMW_TRACE -mllvm -disable-machine-licm -c mw.hip, linked with the regular objects)
for c in ${@:-1 512}; do DHMC_LIB_PATH=$PWD/tools/experiments/_mwtrace/libdhmc_amd.so python tools/experiments/mw_trace.py $c; done
