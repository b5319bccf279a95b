#!/bin/bash
# round-4 GPU call 9: residual-tile prefetch in the 256x320 GEMM main loop -- repeat / parity tests of the RES shapes, same-process A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c9; mkdir -p $O
timeout 600 python -m pytest tests/test_determinism_gpu.py tests/test_duo_gpu.py -q -m gpu -x -k "wide_gemm or linear_repeat or duo_linear" > $O/tests_respf.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests_respf.log
timeout 400 python tools/profile_unet_ops.py --batch 32 --ab "TANGO_RES_PREFETCH=0;TANGO_RES_PREFETCH=1" --rounds 5 --grep "linear" --out $O/res_prefetch_ab_b32.txt > /dev/null 2>$O/ab.err; echo "ab rc=$?"; head -12 $O/res_prefetch_ab_b32.txt | cut -c1-110; grep "linear(wide)" $O/res_prefetch_ab_b32.txt | cut -c1-110
timeout 300 python tools/profile_unet_ops.py --batch 8 --ab "TANGO_RES_PREFETCH=0;TANGO_RES_PREFETCH=1" --rounds 5 --grep "linear" --out $O/res_prefetch_ab_b8.txt > /dev/null 2>$O/ab8.err; echo "ab8 rc=$?"; head -3 $O/res_prefetch_ab_b8.txt | cut -c1-110
