#!/bin/bash
# gpurun call 25 of round 2: smaller batches (B = 8, 16): wide-conv split-K on / off; B = 1 bench (config 2)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2; mkdir -p $O
for b in 8 16; do
timeout 200 python tools/profile_unet_ops.py --batch $b --out $O/unet_ops_b${b}_sk.txt > /dev/null 2>&1; head -1 $O/unet_ops_b${b}_sk.txt
TANGO_NO_WIDE_SPLITK=1 timeout 200 python tools/profile_unet_ops.py --batch $b --out $O/unet_ops_b${b}_nosk.txt > /dev/null 2>&1; head -1 $O/unet_ops_b${b}_nosk.txt
grep "^conv.*splitK" $O/unet_ops_b${b}_sk.txt | head -8; echo; grep "^conv" $O/unet_ops_b${b}_nosk.txt | head -12
done
timeout 600 python bench.py --batch 1 --denoise-steps 100 --no-cpu-baseline > $O/bench_v26_b1.json 2> $O/bench_v26_b1.err; cat $O/bench_v26_b1.json
