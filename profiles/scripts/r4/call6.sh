#!/bin/bash
# round-4 GPU call 6: the whole GPU suite on the tree with schedule 1, the generalised fused cross-attention block, the plan LRU
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c6; mkdir -p $O
timeout 2400 python -m pytest tests/ -q -m gpu -s > $O/tests_gpu.log 2>&1; echo "suite rc=$?"
grep -E "passed|failed" $O/tests_gpu.log | tail -3
grep -E "^FAILED|^ERROR|Error|assert " $O/tests_gpu.log | head -40
