#!/bin/bash
# round-4 GPU call 3: 32x32x16 vs 16x16x32 MFMAs in the 256 x 320 ping-pong loop (tools/loop_probe32.hip), random data
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c3; mkdir -p $O
for shp in "65536 640 11520" "262144 320 5760" "16384 1280 23040" "65536 640 1280"; do
  timeout 120 ./build/loop_probe32 $shp >> $O/loop_probe32.txt 2>&1
done
cat $O/loop_probe32.txt
