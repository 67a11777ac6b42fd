#!/bin/bash
# round 6, second session: where a wave of the fused eta + link kernel spends its clocks (s_memtime around the wait for the staged loads, the
# barrier, the products, the link; printed by a few waves of launches with all 1024 rows active)
O=$PWD/gpurun_out/r6bl; mkdir -p $O
export DHMC_LIB_PATH=$PWD/tools/experiments/_v/lk_clocks/libdhmc_amd.so
timeout -s KILL 200 python bench.py --config 5 --steps 1 --warmup 0 --no-cpu-baseline 2>$O/err.txt | grep -a "LKCLK\|metric" | cut -c1-300 | head -40 | tee $O/clocks.txt
