#!/bin/bash
# round-4 GPU call 10: cooperative single-launch GroupNorm -- parity / repeat / barrier tests, A/B at B = 1 / 8 / 32
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c10; mkdir -p $O
timeout 600 python -m pytest tests/test_gn_coop_gpu.py tests/test_ops_gpu.py -q -m gpu -x -k "gn_coop or groupnorm" > $O/tests_gn.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests_gn.log; grep -E "^FAILED|^ERROR|assert |timed out" $O/tests_gn.log | head
for b in 1 8 32; do
  timeout 300 python tools/profile_unet_ops.py --batch $b --ab "TANGO_NO_GN_COOP=1;TANGO_NO_GN_COOP=0" --rounds 5 --grep "groupnorm" --out $O/gn_coop_ab_b$b.txt > /dev/null 2>$O/ab$b.err; echo "ab b$b rc=$?"; head -3 $O/gn_coop_ab_b$b.txt | cut -c1-110; grep -E "^family groupnorm|^groupnorm" $O/gn_coop_ab_b$b.txt | cut -c1-110
done
