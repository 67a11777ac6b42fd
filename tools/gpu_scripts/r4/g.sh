#!/bin/bash
# round 4: why the v2 kernel waits more — (1) table lookup round trips for a lone wave per SIMD, (2) region-by-region clocks
# of the per-draw kernel, round-3 library against the working tree (tools/experiments/phase_timing.sh)
O=$PWD/gpurun_out/r4g; mkdir -p $O
timeout -s KILL 60 tools/experiments/bin/table_latency | tee $O/table_latency.txt
echo "== v1"; ( cd tools/experiments/_ab/v1 && timeout -s KILL 120 bash tools/experiments/phase_timing.sh 2>&1 | grep -v Warning | tee $O/phase_v1.txt )
echo "== v2"; timeout -s KILL 120 bash tools/experiments/phase_timing.sh 2>&1 | grep -v Warning | tee $O/phase_v2.txt
