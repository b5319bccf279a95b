#!/bin/bash
# round-4 GPU call 11: cooperative GroupNorm, rendezvous memory-ordering variants (TANGO_GN_COOP_MODE 0 / 1 / 2) vs the two-launch path
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c11; mkdir -p $O
timeout 600 python -m pytest tests/test_gn_coop_gpu.py tests/test_ops_gpu.py -q -m gpu -x -k "gn_coop or groupnorm" > $O/tests_gn.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests_gn.log
for b in 1 8 32; do
  timeout 300 python tools/profile_unet_ops.py --batch $b --ab "TANGO_NO_GN_COOP=1;TANGO_GN_COOP_MODE=0;TANGO_GN_COOP_MODE=1;TANGO_GN_COOP_MODE=2" --rounds 5 --grep "groupnorm" --out $O/gn_coop_modes_ab_b$b.txt > /dev/null 2>$O/ab$b.err; echo "ab b$b rc=$?"; head -3 $O/gn_coop_modes_ab_b$b.txt | cut -c1-160; grep -E "^family groupnorm|^groupnorm" $O/gn_coop_modes_ab_b$b.txt | head -9 | cut -c1-160
done
