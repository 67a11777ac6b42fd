#!/bin/bash
O=gpurun_out/r5l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_user_functor.py -q 2>&1 | tail -8 > $O/functor.log; cat $O/functor.log
