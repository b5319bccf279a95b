#!/bin/bash
# round-3 GPU call 7: fused cross-attention block with counted LDS waits -- parity, then read-ahead depth A/B on the B=32 step
O=gpurun_out/r3c7; mkdir -p $O
timeout 600 python -m pytest tests/test_xattn_gpu.py -x -q -s > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
grep -E "passed|failed|rc=|rel err|rel diff|^E  " $O/tests.log | grep -v "print(" | tail -14
for d in 4 8 2; do
  TANGO_EXP_XA_DEPTH=$d timeout 200 python tools/profile_unet_ops.py --batch 32 --out $O/ops_b32_d$d.txt > /dev/null 2>&1
  echo "== depth $d $(head -1 $O/ops_b32_d$d.txt)"; grep -E "xattn" $O/ops_b32_d$d.txt
done
for d in 4 8; do for b in 8 1; do
  TANGO_EXP_XA_DEPTH=$d timeout 200 python tools/profile_unet_ops.py --batch $b --out $O/ops_b${b}_d$d.txt > /dev/null 2>&1
  echo "== B=$b depth $d $(head -1 $O/ops_b${b}_d$d.txt)"; grep -E "xattn" $O/ops_b${b}_d$d.txt
done; done
