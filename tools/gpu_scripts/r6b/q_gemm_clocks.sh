#!/bin/bash
# round 6, second session: the half-batch product's workgroups — when they start and end (100 MHz clock) and where their clocks go — with the two
# half-batches free-running (both multiply at once) and in strict alternation (a product beside the other half's tree kernel)
O=$PWD/gpurun_out/r6bq; mkdir -p $O
export DHMC_LIB_PATH=$PWD/tools/experiments/_v/gemm_clocks/libdhmc_amd.so
for alt in 0 1; do
  DHMC_DENSE="alternate=$alt" timeout -s KILL 200 python bench.py --config 3 --steps 1 --warmup 1 --transitions 10 --no-cpu-baseline 2>$O/err_$alt.txt | grep -a "GEMMCLK" | tail -48 > $O/clocks_alt$alt.txt
  echo "== alternate=$alt"; sort -t' ' -k5,5n $O/clocks_alt$alt.txt | head -48 | cut -c1-200
done
