#!/bin/bash
# PCIe-inclusive rate of config 3 (dense metric, round engine) with host outputs: split into calls of ≈ 1 GiB of draws (default) vs one call
O=gpurun_out/r3aa; mkdir -p $O
cd tools
python pcie_rate.py 100 dense > ../$O/pcie_dense_chunked.json 2>/dev/null
DHMC_HOST_CHUNK=1000000 python pcie_rate.py 100 dense > ../$O/pcie_dense_one_call.json 2>/dev/null
cd ..
cat $O/pcie_dense_chunked.json $O/pcie_dense_one_call.json
