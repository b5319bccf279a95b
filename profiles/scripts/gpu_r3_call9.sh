#!/bin/bash
# round-3 GPU call 9: fused cross-attention block after the prologue / residual rework -- parity, then per-op timing at B = 32 / 8 / 1
O=gpurun_out/r3c9; mkdir -p $O
timeout 600 python -m pytest tests/test_xattn_gpu.py -x -q -s > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
grep -E "passed|failed|rc=|rel err|rel diff|^E  " $O/tests.log | grep -v "print(" | tail -14
for b in 32 8 1; do
  timeout 200 python tools/profile_unet_ops.py --batch $b --out $O/ops_b${b}.txt > /dev/null 2>&1
  echo "== B=$b $(head -1 $O/ops_b$b.txt)"; grep -E "xattn" $O/ops_b$b.txt
done
