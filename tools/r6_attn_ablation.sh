#!/bin/bash
# Round 6: where the time of the S = 4096 self-attention site goes -- parts of attn_kernel<f16, QB = 2, unmasked, MSUM> compiled out one at a
# time (tools/experiments/r6_attention_ablation.patch; results are garbage by construction, only the durations matter), rocprofv3 kernel stats.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-attn_abl}; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for abl in 0 1 2 3 4 5 6; do
  TANGO_ATTN_ABL=$abl timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/a$abl -o st -- python $R/tools/bench_ops.py attention 64 5 4096 8 0 > $OUT/a$abl.log 2>&1
  DB=$(find $OUT/a$abl -name "*.db" | head -1)
  python $R/tools/rocprof_summary.py "$DB" $OUT/a$abl.txt "TANGO_ATTN_ABL=$abl bench_ops.py attention 64 5 4096 8 0" > /dev/null 2>&1
  rm -rf $OUT/a$abl
  echo "== ABL $abl"; grep -E "attn_kernel" $OUT/a$abl.txt | cut -c1-200 | head -2
done
