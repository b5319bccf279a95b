#!/bin/bash
# gpurun call 19 of round 2: q | k | v^T tests; level-0 projection on the wide GEMM A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2; mkdir -p $O
timeout 900 python -m pytest tests/test_determinism_gpu.py -m gpu -q -s -k "qkv" > $O/det_vt.log 2>&1; echo "det rc=$?"; tail -2 $O/det_vt.log; grep -E "^FAILED|rel err" $O/det_vt.log | head -20
timeout 200 python tools/profile_unet_ops.py --out $O/unet_ops_vt.txt > /dev/null 2>&1; head -1 $O/unet_ops_vt.txt; grep "N=960" $O/unet_ops_vt.txt
TANGO_WIDE_VT320=1 timeout 200 python tools/profile_unet_ops.py --out $O/unet_ops_vt320.txt > /dev/null 2>&1; head -1 $O/unet_ops_vt320.txt; grep "N=960" $O/unet_ops_vt320.txt
