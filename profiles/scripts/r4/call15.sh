#!/bin/bash
# round-4 GPU call 15: the whole GPU suite on the tree with gemm_duo routing, cooperative GroupNorm (B = 1), rowvec epilogue, LN xstats, QB2 attention
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c15; mkdir -p $O
timeout 2400 python -m pytest tests/ -q -m gpu -s > $O/tests_gpu.log 2>&1; echo "suite rc=$?"
grep -E "passed|failed" $O/tests_gpu.log | tail -3
grep -E "^FAILED|^ERROR|Error|assert " $O/tests_gpu.log | head -40
