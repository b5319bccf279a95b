#!/bin/bash
# round-4 GPU call 12: the single-key rows' constant in attn1's to_out epilogue (GemmParams::rowvec) -- tests, A/B at B = 32 / 8
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c12; mkdir -p $O
timeout 900 python -m pytest tests/test_single_key_gpu.py tests/test_duo_gpu.py tests/test_gn_coop_gpu.py -q -m gpu -x -s > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|rel diff|separate single" $O/tests.log | tail -5
for b in 32 8; do
  timeout 300 python tools/profile_unet_ops.py --batch $b --ab "TANGO_NO_ROWVEC_FUSE=1;TANGO_NO_ROWVEC_FUSE=0" --rounds 5 --grep "xattn_single|N=320 K=320|N=640 K=640|N=1280 K=1280" --out $O/rowvec_ab_b$b.txt > /dev/null 2>$O/ab$b.err; echo "ab b$b rc=$?"; head -3 $O/rowvec_ab_b$b.txt | cut -c1-110; grep -E "xattn_single|linear" $O/rowvec_ab_b$b.txt | cut -c1-110 | head -12
done
timeout 600 python -m pytest tests/test_parity_batch_gpu.py -q -m gpu -x -s -k "benchmarked_batch and fp16" > $O/parity_batch_fp16.log 2>&1; echo "parity rc=$?"; grep -E "rel err|passed|failed" $O/parity_batch_fp16.log | tail -8
