#!/bin/bash
# round-3 GPU call 8: where does the fused cross-attention kernel spend its time?  Ablation bits (results are wrong with any bit set):
# 1 no DMA waits, 2 no DMA issue in the head loop, 4 no Q / Y fragment streams (LDS reads + MFMA), 8 no epilogue, 16 no barriers in the loop
O=gpurun_out/r3c8; mkdir -p $O
TANGO_EXP_XA_ABL=31 timeout 300 python -m pytest tests/test_xattn_gpu.py -x -q -s -k "fp16 and 4096" 2>&1 | grep -E "abl|passed|failed|rel err" | head
for a in 0 31 4 8; do
  TANGO_EXP_XA_ABL=$a timeout 200 python tools/profile_unet_ops.py --batch 32 --out $O/ops_b32_a$a.txt 2>&1 | grep "xattn abl" | head -2
  echo "abl $a: $(grep -E 'xattn' $O/ops_b32_a$a.txt)"
done
