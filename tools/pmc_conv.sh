#!/bin/bash
# PMC passes over one conv3x3 (implicit GEMM) micro-benchmark; each counter group in its own rocprofv3 run.
# usage: tools/pmc_conv.sh "<bench_ops args>" outdir
cd /tmp; export TMPDIR=/tmp
ARGS="$1"; OUT="$GRAFT_REPO_ROOT/gpurun_out/$2"; mkdir -p "$OUT"
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum" "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "FETCH_SIZE WRITE_SIZE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $grp -d "$OUT/g$i" -o g$i --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_ops.py $ARGS > "$OUT/g$i.log" 2>&1 || echo "group $i failed: $grp"
done
find "$OUT" -name "*counter_collection.csv" | head
