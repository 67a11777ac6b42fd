#!/bin/bash
# round 5: a caller's functor through the pipeline kernel — parity; the rest of the functor suite once (module layout changed)
timeout -s KILL 400 python -m pytest tests/test_gpu_user_functor.py -x -q 2>&1 | tail -8
