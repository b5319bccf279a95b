#!/bin/bash
# round-3 GPU call 11: long-horizon precision ladder (100 steps vs the CPU oracle, 200 steps fp16 vs fp32 engine)
O=gpurun_out/r3c11; mkdir -p $O
TANGO_LONG_TESTS=1 timeout 1100 python -m pytest tests/test_parity_long_gpu.py -x -q -s --durations=3 > $O/ladder.log 2>&1; echo "pytest rc=$?" >> $O/ladder.log
grep -E "passed|failed|rc=|engine vs|^E  |slowest|call " $O/ladder.log | grep -v "print(" | tail -12
