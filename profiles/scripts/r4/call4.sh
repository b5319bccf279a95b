#!/bin/bash
# round-4 GPU call 4: lean multiply part (schedule 2: scalar prep pinned in the read part) under the repeat-run tests, A/B 0 / 1 / 2
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c4; mkdir -p $O
TANGO_WIDE_SCHED=2 timeout 400 python -m pytest tests/test_determinism_gpu.py -m gpu -q -x -k "wide_gemm or conv3x3_wide or linear_qkv_vt" > $O/det_sched2.log 2>&1
echo "sched 2 determinism rc=$?"; tail -2 $O/det_sched2.log; grep -E "rel err|differs" $O/det_sched2.log | head -3
timeout 500 python tools/profile_unet_ops.py --ab "TANGO_WIDE_SCHED=0;TANGO_WIDE_SCHED=1;TANGO_WIDE_SCHED=2" --rounds 3 \
  --grep "conv3x3|linear" --out $O/ab_sched_b32.txt > /dev/null 2> $O/ab_err.log; echo "ab rc=$?"; head -14 $O/ab_sched_b32.txt
timeout 300 python tools/profile_unet_ops.py --batch 8 --ab "TANGO_WIDE_SCHED=1;TANGO_WIDE_SCHED=2" --rounds 3 \
  --grep "conv3x3|linear" --out $O/ab_sched_b8.txt > /dev/null 2>> $O/ab_err.log; echo "ab8 rc=$?"; head -8 $O/ab_sched_b8.txt
