#!/bin/bash
# round-4 GPU call 14: short-sequence attention with 32 query rows per wave at large batch; one-launch GroupNorm size threshold
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c14; mkdir -p $O
for b in 32 8; do
  timeout 300 python tools/profile_unet_ops.py --batch $b --ab "TANGO_ATTN_QB2_MIN_WGS=0;TANGO_ATTN_QB2_MIN_WGS=512;TANGO_ATTN_QB2_MIN_WGS=1024;TANGO_ATTN_QB2_MIN_WGS=2048" --rounds 5 --grep "attention" --out $O/attn_qb2_ab_b$b.txt > /dev/null 2>$O/ab$b.err; echo "attn ab b$b rc=$?"; head -3 $O/attn_qb2_ab_b$b.txt | cut -c1-160; grep -E "^attention|family attention" $O/attn_qb2_ab_b$b.txt | cut -c1-160
  timeout 300 python tools/profile_unet_ops.py --batch $b --ab "TANGO_GN_SMALL_MB=8;TANGO_GN_SMALL_MB=16;TANGO_GN_SMALL_MB=48" --rounds 5 --grep "groupnorm" --out $O/gn_small_ab_b$b.txt > /dev/null 2>$O/abg$b.err; echo "gn ab b$b rc=$?"; grep -E "^TOTAL|family groupnorm|rows=64 |rows=256 " $O/gn_small_ab_b$b.txt | cut -c1-135
done
timeout 300 env TANGO_ATTN_QB2_MIN_WGS=512 python -m pytest tests/test_ops_gpu.py tests/test_determinism_gpu.py -q -m gpu -x -k "attention" > $O/tests_attn.log 2>&1; echo "attn tests (QB2 forced) rc=$?"; tail -2 $O/tests_attn.log
