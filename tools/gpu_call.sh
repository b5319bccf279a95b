#!/bin/bash
# gpurun call 31 of round 2: rocprofv3 kernel stats of the final tree + one more bench sample (no CPU leg)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/final; mkdir -p $OUT
timeout 300 python bench.py --no-cpu-baseline > $R/gpurun_out/r2/bench_v27_nocpu.json 2> $R/gpurun_out/r2/bench_v27_nocpu.err; cut -c1-700 $R/gpurun_out/r2/bench_v27_nocpu.json
cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --batch 32 --denoise-steps 4 --steps 1 --warmup 1 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o st -- $CMD > $OUT/stats.log 2>&1
DB=$(find $OUT/stats -name "*.db" | head -1)
python $R/tools/rocprof_summary.py "$DB" $OUT/kernel_stats.txt "python bench.py --batch 32 --denoise-steps 4 --steps 1 --warmup 1 --no-cpu-baseline"
rm -rf $OUT/stats
head -16 $OUT/kernel_stats.txt | cut -c1-170
