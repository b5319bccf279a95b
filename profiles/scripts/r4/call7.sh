#!/bin/bash
# round-4 GPU call 7: gemm_duo.hip (256 x 160, two workgroups per CU) -- parity / repeat tests, then same-process A/Bs of the K bound
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c7; mkdir -p $O
timeout 600 python -m pytest tests/test_duo_gpu.py -q -m gpu -x > $O/tests_duo.log 2>&1; echo "duo tests rc=$?"; tail -3 $O/tests_duo.log
grep -E "^FAILED|^ERROR|assert |rel err" $O/tests_duo.log | head -20
AB="TANGO_DUO_MAXK=0;TANGO_DUO_MAXK=320;TANGO_DUO_MAXK=640;TANGO_DUO_MAXK=1280;TANGO_DUO_MAXK=2560;TANGO_DUO_MAXK=5120"
timeout 500 python tools/profile_unet_ops.py --batch 32 --ab "$AB" --rounds 3 --grep "linear" --out $O/duo_maxk_ab_b32.txt > /dev/null 2>$O/ab32.err; echo "ab32 rc=$?"; head -12 $O/duo_maxk_ab_b32.txt
AB2="TANGO_DUO_MAXK=1280,TANGO_DUO_MASK=7;TANGO_DUO_MAXK=1280,TANGO_DUO_MASK=7,TANGO_DUO_PRIO=1;TANGO_DUO_MAXK=1280,TANGO_DUO_MASK=15;TANGO_DUO_MAXK=1280,TANGO_DUO_MASK=7,TANGO_NO_STREAM_LN_GEGLU=1;TANGO_DUO_MAXK=0,TANGO_NO_STREAM_LN_GEGLU=1;TANGO_DUO_MAXK=1280,TANGO_DUO_MASK=7,TANGO_DUO_MIN_TILES=256"
timeout 500 python tools/profile_unet_ops.py --batch 32 --ab "$AB2" --rounds 3 --grep "linear|layernorm" --out $O/duo_variants_ab_b32.txt > /dev/null 2>$O/ab32v.err; echo "ab32v rc=$?"; head -8 $O/duo_variants_ab_b32.txt
AB8="TANGO_DUO_MAXK=0;TANGO_DUO_MAXK=640,TANGO_DUO_MIN_TILES=256;TANGO_DUO_MAXK=1280,TANGO_DUO_MIN_TILES=256;TANGO_DUO_MAXK=5120,TANGO_DUO_MIN_TILES=256;TANGO_DUO_MAXK=5120,TANGO_DUO_MIN_TILES=128;TANGO_DUO_MAXK=5120,TANGO_DUO_MIN_TILES=512"
timeout 400 python tools/profile_unet_ops.py --batch 8 --ab "$AB8" --rounds 3 --grep "linear" --out $O/duo_ab_b8.txt > /dev/null 2>$O/ab8.err; echo "ab8 rc=$?"; head -8 $O/duo_ab_b8.txt
