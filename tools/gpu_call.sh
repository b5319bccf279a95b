#!/bin/bash
# gpurun call 28 of round 2: GroupNorm with four rows in flight per thread -- tests, A/B against the previous library
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "groupnorm or norm" > $O/ops_gn.log 2>&1; echo "ops rc=$?"; tail -1 $O/ops_gn.log
timeout 200 python tools/profile_unet_ops.py --out $O/unet_ops_gn4.txt > /dev/null 2>&1; head -1 $O/unet_ops_gn4.txt; grep "^groupnorm" $O/unet_ops_gn4.txt | head -8
cp tango_amd/lib/libtango_hip.so /tmp/new.so; cp build/libtango_hip_gnbase.so tango_amd/lib/libtango_hip.so
timeout 200 python tools/profile_unet_ops.py --out $O/unet_ops_gn1.txt > /dev/null 2>&1; head -1 $O/unet_ops_gn1.txt; grep "^groupnorm" $O/unet_ops_gn1.txt | head -8
cp /tmp/new.so tango_amd/lib/libtango_hip.so
