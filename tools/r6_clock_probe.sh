#!/bin/bash
# Round 6: shader clock and socket power while the level-0 wide conv runs back to back on random vs zero-filled operands (rocm-smi
# sampled from a second process): the evidence that the conv main loop is DVFS-bound on real data.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-clock}; mkdir -p $OUT
for fill in randn zeros; do
  python $R/tools/bench_ops.py conv 64 320 256 16 320 1500 $fill > $OUT/run_$fill.log 2>&1 &
  PID=$!
  sleep 6          # import + first launches
  echo "== fill $fill" >> $OUT/clock_power.txt
  for i in 1 2 3 4 5 6; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|fclk|mclk" | tr -s ' ' | head -6 >> $OUT/clock_power.txt
    echo "--" >> $OUT/clock_power.txt
    sleep 1
  done
  wait $PID
done
rocm-smi --showmaxpower 2>/dev/null | grep -i power >> $OUT/clock_power.txt
cat $OUT/clock_power.txt
