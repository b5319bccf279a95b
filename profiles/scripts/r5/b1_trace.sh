# kernel stats and kernel-to-kernel gaps of the B = 1 bench command on the final tree (same command as profiles/r4_c8_kernel_stats_b1.txt)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/b1trace; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --batch 1 --denoise-steps 20 --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/tr -o st -- $CMD > $O/tr.log 2>&1
DB=$(find $O/tr -name "*.db" | head -1)
python $R/tools/kernel_gaps.py "$DB" $O/kernel_gaps_b1.txt "$CMD"; python $R/tools/rocprof_summary.py "$DB" $O/kernel_stats_b1.txt "$CMD"
rm -rf $O/tr; head -4 $O/kernel_gaps_b1.txt | cut -c1-200; head -12 $O/kernel_stats_b1.txt | cut -c1-170
