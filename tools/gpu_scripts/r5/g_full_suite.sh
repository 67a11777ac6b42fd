#!/bin/bash
# round 5: the whole GPU suite (packed engine is the default for small diagonal-metric chains), with durations
O=gpurun_out/r5g; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=15 2>&1 | tail -60 > $O/pytest.log; cat $O/pytest.log
