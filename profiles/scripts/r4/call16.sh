#!/bin/bash
# round-4 GPU call 16: persistent 256x320 GEMM (next tile's first chunk prefetched behind the epilogue) -- bit-equality tests, A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c16; mkdir -p $O
timeout 900 python -m pytest tests/test_duo_gpu.py -q -m gpu -x -k "pers" > $O/tests_pers.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests_pers.log; grep -E "^FAILED|^ERROR|assert |differ" $O/tests_pers.log | head
for b in 32 8; do
  timeout 400 python tools/profile_unet_ops.py --batch $b --ab "TANGO_WIDE_PERS=0;TANGO_WIDE_PERS=1;TANGO_WIDE_PERS=2;TANGO_WIDE_PERS=4" --rounds 5 --grep "linear" --out $O/wide_pers_ab_b$b.txt > /dev/null 2>$O/ab$b.err; echo "ab b$b rc=$?"; head -3 $O/wide_pers_ab_b$b.txt | cut -c1-160; grep -E "^family linear|^linear" $O/wide_pers_ab_b$b.txt | cut -c1-160 | head -26
done
