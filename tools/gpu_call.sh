#!/bin/bash
# gpurun call 30 of round 2: final validation of the tree (files not re-run since the last kernel changes)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_full_gpu.py -m gpu -q -s -k "fp16 or bf16 or xl or XL" > $O/final_parity.log 2>&1; echo "parity rc=$?"; tail -2 $O/final_parity.log; grep -E "engine vs|rel err" $O/final_parity.log
timeout 900 python -m pytest tests/test_determinism_gpu.py tests/test_string_ckpt_gpu.py -m gpu -q > $O/final_det.log 2>&1; echo "det+string rc=$?"; tail -2 $O/final_det.log
