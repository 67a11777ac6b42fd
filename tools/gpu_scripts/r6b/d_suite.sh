#!/bin/bash
# round 6, second session: the whole GPU suite and the fuzz sweep on the build with the shared-reciprocal link and the rotated product kernel
O=$PWD/gpurun_out/r6bd; mkdir -p $O
timeout -s KILL 1200 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -30 > $O/pytest.log; tail -4 $O/pytest.log
timeout -s KILL 300 python tools/fuzz_parity.py 60 20261001 2>/dev/null | tail -2 > $O/fuzz.txt; cat $O/fuzz.txt
