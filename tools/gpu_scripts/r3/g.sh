#!/bin/bash
# the trajectory-ends-in-LDS layout forced on for EVERY family (library built with -DDHMC_FORCE_TRAJ_LDS): whole GPU suite + fuzz
set -u
O=gpurun_out/r3g; mkdir -p $O
export PYTHONPATH=tests
export DHMC_LIB_PATH=$PWD/tools/experiments/_v/all_tpl/libdhmc_amd.so
timeout -s KILL 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_parity.py::test_host_outputs_in_chunks_equal_device_outputs > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee $O/log.txt
tail -8 $O/pytest.txt | tee -a $O/log.txt
timeout -s KILL 200 python tools/fuzz_parity.py 120 7 > $O/fuzz.txt 2>&1
echo "fuzz rc=$?" | tee -a $O/log.txt
tail -5 $O/fuzz.txt | tee -a $O/log.txt
