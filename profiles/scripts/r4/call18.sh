#!/bin/bash
# round-4 GPU call 18: level-0 GEGLU projection -- streaming kernel (folded LN, in-loop statistics) vs statistics pass + persistent 256x320 GEMM
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c18; mkdir -p $O
for b in 32 8; do
  timeout 300 python tools/profile_unet_ops.py --batch $b --ab "TANGO_NO_STREAM_LN_GEGLU=0;TANGO_NO_STREAM_LN_GEGLU=1" --rounds 5 --grep "N=2560 K=320|ln_stats|layernorm" --out $O/l0_geglu_route_ab_b$b.txt > /dev/null 2>$O/ab$b.err; echo "ab b$b rc=$?"; head -3 $O/l0_geglu_route_ab_b$b.txt | cut -c1-110; grep -E "N=2560|ln_stats|layernorm" $O/l0_geglu_route_ab_b$b.txt | cut -c1-110
done
