#!/bin/bash
# PMC passes over the two activation-stationary kernels of round 6 (ff_fused_kernel, qkv_stat_kernel) at the benchmarked size, each counter
# group in its own rocprofv3 run (no trace domains beside --kernel-trace).  Output: gpurun_out/r6_pmc_stat/summary.txt
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r6_pmc_stat; mkdir -p $OUT
run() {  # name, kernel-name substring, driver args
  local name=$1 kern=$2; shift 2
  local i=0
  for grp in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 180 rocprofv3 --kernel-trace --pmc $grp -d $OUT/${name}_g$i -o g --output-format csv -- python $R/tools/ff_fused_ablation.py "$@" > $OUT/${name}_g$i.log 2>&1 || echo "$name group $i failed"
  done
  python - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob("$OUT/${name}_g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$kern" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
with open("$OUT/summary.txt", "a") as o:
    o.write("## $name  (tools/ff_fused_ablation.py $*; kernel *$kern*)\n")
    for k in sorted(tot):
        o.write("%-32s mean_per_launch %.4g launches %d\n" % (k, tot[k] / cnt[k], cnt[k]))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in tot and "SQ_BUSY_CU_CYCLES" in tot:
        o.write("derived: MFMA busy %.1f %% of (4 SIMDs x busy-CU cycles); wave cycles: waiting %.1f %%, issue-stalled %.1f %%, issuing %.1f %%\n" % (
            100 * tot["SQ_VALU_MFMA_BUSY_CYCLES"] / cnt["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * tot["SQ_BUSY_CU_CYCLES"] / cnt["SQ_BUSY_CU_CYCLES"]),
            100 * tot["SQ_WAIT_ANY"] / tot["SQ_WAVE_CYCLES"], 100 * tot["SQ_WAIT_INST_ANY"] / tot["SQ_WAVE_CYCLES"],
            100 * (tot["SQ_ACTIVE_INST_ANY"] / cnt["SQ_ACTIVE_INST_ANY"]) / (tot["SQ_WAVE_CYCLES"] / cnt["SQ_WAVE_CYCLES"])))
    if "FETCH_SIZE" in tot:
        o.write("derived: FETCH_SIZE x 2 (gfx950 correction, KiB) = %.1f MB, WRITE_SIZE = %.1f MB per launch\n" % (
            2 * tot["FETCH_SIZE"] / cnt["FETCH_SIZE"] * 1024 / 1e6, tot.get("WRITE_SIZE", 0.0) / max(cnt.get("WRITE_SIZE", 1), 1) * 1024 / 1e6))
    o.write("\n")
PY
  rm -rf $OUT/${name}_g*/
}
rm -f $OUT/summary.txt
run ff_fused_m262144 ff_fused_kernel one 0
run two_gemm_route_geglu_stream lin_stream_kernel one 1
run qkv_stat_m262144 qkv_stat_kernel qkv 0
cat $OUT/summary.txt
