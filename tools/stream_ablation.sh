#!/bin/bash
# Ablation timing of the streaming linear kernel (TANGO_STREAM_ABL bits: 1 no residual loads, 2 no output stores, 4 no activation
# ring refills); needs a build with the ablation hooks.  usage: tools/stream_ablation.sh "<bench_ops linear args>" "<abl list>"
cd /tmp; export TMPDIR=/tmp
for abl in $2; do
  rm -rf /tmp/sabl_$abl
  TANGO_STREAM_ABL=$abl timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/sabl_$abl -o a --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_ops.py $1 > /dev/null 2>&1
  echo "abl=$abl $(grep -E 'lin_stream' /tmp/sabl_$abl/a_kernel_stats.csv | awk -F, '{print "avg_ns=" $4}')"
done
