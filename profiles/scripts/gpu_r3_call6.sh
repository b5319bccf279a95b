#!/bin/bash
O=gpurun_out/r3c6; mkdir -p $O
timeout 600 python -m pytest tests/test_xattn_gpu.py -q -s > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
grep -E "passed|failed|rc=|rel err|rel diff|xattn block|differs|AssertionError" $O/tests.log | grep -v "print(\|raise" | tail -30
