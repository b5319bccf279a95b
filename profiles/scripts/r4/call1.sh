#!/bin/bash
# round-4 GPU call 1: (a) the merged single-key CFG path, (b) the three new DMA-issue schedules of the 256 x 320 kernels under the
# parity + repeat-run tests of those kernels, (c) same-process A/B of the four schedules on one UNet step at B = 32
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c1; mkdir -p $O
timeout 400 python -m pytest tests/test_single_key_gpu.py -m gpu -q -x > $O/single_key.log 2>&1; echo "single_key rc=$?"; tail -3 $O/single_key.log
for S in 2 1 3; do
  TANGO_WIDE_SCHED=$S timeout 400 python -m pytest tests/test_determinism_gpu.py -m gpu -q -x -k "wide_gemm or conv3x3_wide" > $O/det_sched$S.log 2>&1
  echo "sched $S determinism rc=$?"; tail -2 $O/det_sched$S.log; grep -E "rel err|differs" $O/det_sched$S.log | head -3
done
timeout 500 python tools/profile_unet_ops.py --ab "TANGO_WIDE_SCHED=0;TANGO_WIDE_SCHED=1;TANGO_WIDE_SCHED=2;TANGO_WIDE_SCHED=3" --rounds 3 \
  --grep "conv3x3|linear" --out $O/ab_sched_b32.txt > /dev/null 2> $O/ab_err.log; echo "ab rc=$?"; head -12 $O/ab_sched_b32.txt
