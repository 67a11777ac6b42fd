#!/bin/bash
export PYTHONPATH=tests
O=gpurun_out/r3m; mkdir -p $O
timeout -s KILL 600 python -m pytest tests/test_gpu_configs.py -m gpu -q -k "bench_launches" > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee $O/log.txt; tail -6 $O/pytest.txt | tee -a $O/log.txt
timeout -s KILL 600 python bench.py --gpus 2 --chains 2048 --steps 10 --warmup 2 --no-cpu-baseline 2> $O/two_rank.err | tail -1 > $O/two_rank_bench.json
cut -c1-700 $O/two_rank_bench.json | tee -a $O/log.txt
