#!/bin/bash
# round 6, second session: gemm_rows_f64_kernel with the A tile's LDS columns rotated per k-group (no 4-way conflict on the transposing stores)
# and 16-byte B stores: parity of every MFMA path, configs 3 and 5
O=gpurun_out/r6bc; mkdir -p $O
timeout -s KILL 1500 python -m pytest tests/test_gpu_dense.py tests/test_gpu_configs.py tests/test_gpu_engines.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt
for c in 3 5; do
 for i in 1 2; do
  r=$(timeout -s KILL 400 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline 2>$O/err_c$c.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g leapfrog-steps/s, %.1f ms/step, frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))")
  echo "c$c: $r" | tee -a $O/bench.txt
 done
done
