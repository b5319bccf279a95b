#!/bin/bash
# Round 6: (a) does the 512 x 160 form of the conv tile run where the switch says so (kernel names), (b) is the wide conv in the
# power-limited regime (zero-filled vs random operands, same kernel), per-kernel durations from rocprofv3 --kernel-trace --stats.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-regime}; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
run() {  # name, env..., -- args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/$name -o st -- python $R/tools/bench_ops.py "$@" > $OUT/$name.log 2>&1
  DB=$(find $OUT/$name -name "*.db" | head -1)
  python $R/tools/rocprof_summary.py "$DB" $OUT/$name.txt "${envs[*]} bench_ops.py $*" > /dev/null 2>&1
  rm -rf $OUT/$name
  echo "== $name"; grep -E "conv3x3|gemm_wide" $OUT/$name.txt | cut -c1-200 | head -4
}
# level-0 ResBlock conv at config-3 size (B2 = 64) and the level-1 conv
for fill in randn zeros ones; do
  run l0_wide_$fill TANGO_CONV_TALL=0 -- conv 64 320 256 16 320 12 $fill
  run l0_tall_$fill TANGO_CONV_TALL=1 -- conv 64 320 256 16 320 12 $fill
done
run l1_wide_randn TANGO_CONV_TALL=0 -- conv 64 640 128 8 640 12 randn
run l1_tall_randn TANGO_CONV_TALL=1 -- conv 64 640 128 8 640 12 randn
run l1_wide_zeros TANGO_CONV_TALL=0 -- conv 64 640 128 8 640 12 zeros
