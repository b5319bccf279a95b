#!/bin/bash
# round-4 GPU call 5: wave-priority modes of the 256 x 320 ping-pong loops (TANGO_WIDE_PRIO 0 / 1 / 2) on schedule 1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c5; mkdir -p $O
for P in 0 2; do
TANGO_WIDE_PRIO=$P timeout 400 python -m pytest tests/test_determinism_gpu.py -m gpu -q -x -k "wide_gemm or conv3x3_wide or linear_qkv_vt" > $O/det_prio$P.log 2>&1
echo "prio $P determinism rc=$?"; tail -2 $O/det_prio$P.log; grep -E "rel err|differs" $O/det_prio$P.log | head -3
done
timeout 500 python tools/profile_unet_ops.py --ab "TANGO_WIDE_PRIO=0;TANGO_WIDE_PRIO=1;TANGO_WIDE_PRIO=2" --rounds 3 \
  --grep "conv3x3|linear" --out $O/ab_prio_b32.txt > /dev/null 2> $O/ab_err.log; echo "ab rc=$?"; head -14 $O/ab_prio_b32.txt
