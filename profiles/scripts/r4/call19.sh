#!/bin/bash
# round-4 GPU call 19: PMC view (MFMA busy, wave wait / stall fractions, LDS, HBM-side bytes) of the round's GEMM kernels on the final tree
cd $GRAFT_REPO_ROOT
export PMC_OUT=r4c19
ALL=$GRAFT_REPO_ROOT/gpurun_out/r4c19/pmc_gemm_kernels_summary.txt; mkdir -p gpurun_out/r4c19; rm -f $ALL
one() { bash tools/pmc_op.sh run "$@" > /dev/null 2>&1; cat gpurun_out/r4c19/summary.txt >> $ALL; }
one gemm_wide_pers_geglu_l1 gemm_wide_pers linear 65536 5120 640 3 nores geglu
one gemm_wide_pers_res_l0 gemm_wide_pers linear 262144 320 1280 3 res
one gemm_wide_oneshot_res_l2 gemm_wide_kernel linear 16384 1280 1280 3 res
one gemm_duo_res_half_l1 gemm_duo linear 32768 640 640 3 res
grep -E "^##|derived" $ALL
