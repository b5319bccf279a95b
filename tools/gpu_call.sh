#!/bin/bash
# gpurun call 22 of round 2: staggered starts on the wide GEMM (residual-epilogue shapes)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2; mkdir -p $O
export TANGO_WIDE_TRACE=1
for st in 0 1600 3200 6400; do
echo "stagger $st (10 ns)"
{
TANGO_WIDE_STAGGER=$st python tools/bench_ops.py linear 262144 320 640 3 res
TANGO_WIDE_STAGGER=$st python tools/bench_ops.py linear 262144 320 1280 3 res
TANGO_WIDE_STAGGER=$st python tools/bench_ops.py linear 65536 640 640 3 res
TANGO_WIDE_STAGGER=$st python tools/bench_ops.py linear 65536 5120 640 3 nores geglu
} 2>&1 | grep trace | awk 'NR%3==0' | sed 's/gemm_wide trace //'
done | tee $O/wide_stagger.txt
