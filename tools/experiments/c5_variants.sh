#!/bin/bash
ROOT=$(cd $(dirname $0)/../.. && pwd)
python $ROOT/tools/experiments/c5_rounds.py 10 | head -1
for so in $ROOT/tools/experiments/_v/*/libdhmc_amd.so; do echo $(basename $(dirname $so)); DHMC_LIB_PATH=$so python $ROOT/tools/experiments/c5_rounds.py 10 | head -1; done
