#!/bin/bash
# Round 6: PMC of the two kernels added at the end of the round -- gn_slab_kernel (UNet level 2 / 3 GroupNorm shapes at B2 = 64) and the 256 x 32 tile of
# conv3x3_halo_kernel on the UNet's conv_out (320 -> 8 channels) -- through tools/pmc_op.sh (each counter group in its own rocprofv3 --kernel-trace --pmc run).
R=$GRAFT_REPO_ROOT
export PMC_OUT=r6_pmc_slab
rm -f $R/gpurun_out/$PMC_OUT/summary_all.txt; mkdir -p $R/gpurun_out/$PMC_OUT
bash $R/tools/pmc_op.sh run gn_slab_c1280_rows256 gn_slab groupnorm 64 1280 256 5 1 > /dev/null 2>&1; cat $R/gpurun_out/$PMC_OUT/summary.txt >> $R/gpurun_out/$PMC_OUT/summary_all.txt
bash $R/tools/pmc_op.sh run gn_slab_c1280_rows64 gn_slab groupnorm 64 1280 64 5 1 > /dev/null 2>&1; cat $R/gpurun_out/$PMC_OUT/summary.txt >> $R/gpurun_out/$PMC_OUT/summary_all.txt
bash $R/tools/pmc_op.sh run conv_out_halo_narrow conv3x3_halo conv 64 320 256 16 8 5 > /dev/null 2>&1; cat $R/gpurun_out/$PMC_OUT/summary.txt >> $R/gpurun_out/$PMC_OUT/summary_all.txt
cat $R/gpurun_out/$PMC_OUT/summary_all.txt
