#!/bin/bash
# Ablation timing of the halo conv kernel (TANGO_HALO_ABL bits: 1 no weight DMA, 2 no halo DMA, 4 no MFMA/ds_read, 8 no ds_read)
# usage: tools/halo_ablation.sh "<bench_ops conv args>" "<abl list>"
cd /tmp; export TMPDIR=/tmp
for abl in $2; do
  rm -rf /tmp/abl_$abl
  TANGO_HALO_ABL=$abl timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/abl_$abl -o a --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_ops.py $1 > /dev/null 2>&1
  echo "abl=$abl $(grep -E 'conv3x3_halo|gemm_dma' /tmp/abl_$abl/a_kernel_stats.csv | awk -F, '{print $1, "avg_ns=" $4}' | cut -c1-120)"
done
