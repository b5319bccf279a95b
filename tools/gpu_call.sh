#!/bin/bash
# gpurun call 29 of round 2: short-K linears on the wide GEMM
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2; mkdir -p $O
timeout 600 python -m pytest tests/test_determinism_gpu.py -m gpu -q -x -k "wide_gemm_repeat" > $O/det_shortk.log 2>&1; echo "det rc=$?"; tail -2 $O/det_shortk.log; grep -E "rel err|differs" $O/det_shortk.log | head -5
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -m gpu -q -x > $O/ops_shortk.log 2>&1; echo "ops+engine rc=$?"; tail -1 $O/ops_shortk.log
timeout 200 python tools/profile_unet_ops.py --out $O/unet_ops_v27.txt > /dev/null 2>&1; head -1 $O/unet_ops_v27.txt; grep "K=96" $O/unet_ops_v27.txt
