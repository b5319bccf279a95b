#!/bin/bash
# gpurun call 35 of round 2: level-3 convs (four 32 x 2 images per tile, halo 544) on the wide conv with split-K
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2; mkdir -p $O
timeout 300 python -m pytest tests/test_determinism_gpu.py -m gpu -q -x -k "conv3x3_wide" > $O/det_l3.log 2>&1; echo "det rc=$?"; tail -2 $O/det_l3.log; grep -E "rel err|differs" $O/det_l3.log | head -3
timeout 200 python tools/profile_unet_ops.py --out $O/unet_ops_l3.txt > /dev/null 2>&1; head -1 $O/unet_ops_l3.txt; grep "M=4096.*splitK" $O/unet_ops_l3.txt
TANGO_NO_WIDE_SPLITK=1 timeout 200 python tools/profile_unet_ops.py --out $O/unet_ops_nol3.txt > /dev/null 2>&1; head -1 $O/unet_ops_nol3.txt; grep "M=4096.*splitK" $O/unet_ops_nol3.txt
