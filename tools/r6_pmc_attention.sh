#!/bin/bash
# Round 6: PMC of the S = 4096 self-attention site (B2 = 64, 5 heads, fp16) in its three forms -- register-staged tiles, K by LDS-DMA, K and V^T by LDS-DMA
# (tango_op_attention_ex flags bit 2) -- through tools/pmc_op.sh (each counter group in its own rocprofv3 --kernel-trace --pmc run).
R=$GRAFT_REPO_ROOT
export PMC_OUT=r6_pmc_attn
rm -f $R/gpurun_out/$PMC_OUT/summary_all.txt; mkdir -p $R/gpurun_out/$PMC_OUT
TANGO_ATTN_KDMA=0 bash $R/tools/pmc_op.sh run attn_s4096_staged attn_kernel attention 64 5 4096 3 0 > /dev/null 2>&1; cat $R/gpurun_out/$PMC_OUT/summary.txt >> $R/gpurun_out/$PMC_OUT/summary_all.txt
bash $R/tools/pmc_op.sh run attn_s4096_kdma attn_kernel attention 64 5 4096 3 0 > /dev/null 2>&1; cat $R/gpurun_out/$PMC_OUT/summary.txt >> $R/gpurun_out/$PMC_OUT/summary_all.txt
bash $R/tools/pmc_op.sh run attn_s4096_kvdma attn_kernel attention 64 5 4096 3 4 > /dev/null 2>&1; cat $R/gpurun_out/$PMC_OUT/summary.txt >> $R/gpurun_out/$PMC_OUT/summary_all.txt
cat $R/gpurun_out/$PMC_OUT/summary_all.txt
