#!/bin/bash
# PMC passes (SQ busy / wait, MFMA busy, LDS, HBM-side bytes) over single-op micro-benchmarks (round 4 copy of tools/pmc_kernels.sh:
# output dir from $PMC_OUT, and "pmc.sh run <name> <kernel substring> <bench_ops args>" measures one op); each counter
# group in its own rocprofv3 run (no trace domains beside --kernel-trace).  usage: tools/pmc_kernels.sh ; output gpurun_out/${PMC_OUT:-r4/pmc}/
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${PMC_OUT:-r4/pmc}; mkdir -p $OUT
run() {  # name, kernel-name substring, bench_ops args
  local name=$1 kern=$2; shift 2
  local i=0
  for grp in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 120 rocprofv3 --kernel-trace --pmc $grp -d $OUT/${name}_g$i -o g --output-format csv -- python $R/tools/bench_ops.py "$@" > $OUT/${name}_g$i.log 2>&1 || echo "$name group $i failed"
  done
  python - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob("$OUT/${name}_g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$kern" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
with open("$OUT/summary.txt", "a") as o:
    o.write("## $name  (bench_ops.py $*; kernel *$kern*)\n")
    for k in sorted(tot):
        o.write("%-32s mean_per_launch %.4g launches %d\n" % (k, tot[k] / cnt[k], cnt[k]))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in tot and "SQ_BUSY_CU_CYCLES" in tot:
        o.write("derived: MFMA busy %.1f %% of (4 SIMDs x busy-CU cycles); wave cycles: waiting %.1f %%, issue-stalled %.1f %%, issuing %.1f %%\n" % (
            100 * tot["SQ_VALU_MFMA_BUSY_CYCLES"] / cnt["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * tot["SQ_BUSY_CU_CYCLES"] / cnt["SQ_BUSY_CU_CYCLES"]),
            100 * tot["SQ_WAIT_ANY"] / tot["SQ_WAVE_CYCLES"], 100 * tot["SQ_WAIT_INST_ANY"] / tot["SQ_WAVE_CYCLES"],
            100 * (tot["SQ_ACTIVE_INST_ANY"] / cnt["SQ_ACTIVE_INST_ANY"]) / (tot["SQ_WAVE_CYCLES"] / cnt["SQ_WAVE_CYCLES"])))
    o.write("\n")
PY
  rm -rf $OUT/${name}_g*/
}
rm -f $OUT/summary.txt
if [ $# -gt 0 ]; then "$@"; cat $OUT/summary.txt; exit 0; fi
run conv_wide_l0 conv3x3_wide conv 64 320 256 16 320 3
run gemm_wide_geglu_l1 gemm_wide linear 65536 5120 640 3 nores geglu
run gemm_wide_res_l0 gemm_wide linear 262144 320 1280 3 res
run stream_n320_k320 lin_stream linear 262144 320 320 3 res
cat $OUT/summary.txt

