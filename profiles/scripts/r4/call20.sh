#!/bin/bash
# round-4 GPU call 20: long-horizon precision ladder on the final tree (100 DDPM steps vs the CPU oracle at B = 1; 200 steps fp16 vs fp32 engine)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c20; mkdir -p $O
TANGO_LONG_TESTS=1 timeout 700 python -m pytest tests/test_parity_long_gpu.py -q -m gpu -s --durations=3 > $O/long_horizon_ladder.log 2>&1; echo "pytest rc=$?" >> $O/long_horizon_ladder.log
grep -E "DDPM steps|passed|failed|rc=" $O/long_horizon_ladder.log
