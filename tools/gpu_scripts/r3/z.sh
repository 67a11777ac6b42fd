#!/bin/bash
# config 3 with the square products through the v2 GEMM variants (tools/experiments/_v/gemmv2_*), 2 parts and 1 part
for parts in 2 1; do
  echo "DHMC_DENSE_PARTS=$parts"
  DHMC_DENSE_PARTS=$parts bash tools/experiments/time_variants.sh --config 3 --transitions 20
done
