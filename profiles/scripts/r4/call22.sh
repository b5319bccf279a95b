#!/bin/bash
# round-4 GPU call 22: PMC view of the attention kernel (unmasked, 32 query rows per wave, matrix-pipe row sums) inside a 2-step denoise at B = 32
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4c22; mkdir -p $OUT
KERN="attn_kernelIDF16_Li2ELb0ELi4ELi3ELb0ELb0ELb1"
CMD="python $R/bench.py --batch 32 --denoise-steps 2 --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs"
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $grp -d $OUT/g$i -o g --output-format csv -- $CMD > $OUT/g$i.log 2>&1 || echo "group $i failed"
done
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$KERN" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
with open("$OUT/pmc_attention_summary.txt", "w") as o:
    o.write("## attention kernel *$KERN* (S = 4096 and S = 1024 self-attention sites of a 2-step denoise at B = 32; $CMD)\n")
    for k in sorted(tot):
        o.write("%-32s mean_per_launch %.4g launches %d\n" % (k, tot[k] / cnt[k], cnt[k]))
    m = lambda k: tot[k] / max(cnt[k], 1)
    if cnt["SQ_BUSY_CU_CYCLES"]:
        o.write("derived: MFMA busy %.1f %% of (4 SIMDs x busy-CU cycles); wave cycles: waiting %.1f %%, issue-stalled %.1f %%, issuing %.1f %%\n" % (
            100 * m("SQ_VALU_MFMA_BUSY_CYCLES") / (4 * m("SQ_BUSY_CU_CYCLES")), 100 * m("SQ_WAIT_ANY") / m("SQ_WAVE_CYCLES"),
            100 * m("SQ_WAIT_INST_ANY") / m("SQ_WAVE_CYCLES"), 100 * m("SQ_ACTIVE_INST_ANY") / m("SQ_WAVE_CYCLES")))
        if cnt["SQ_ACTIVE_INST_VALU"]:
            o.write("derived: VALU active %.1f %% of wave cycles; LDS active %.1f %%\n" % (100 * m("SQ_ACTIVE_INST_VALU") / m("SQ_WAVE_CYCLES"), 100 * m("SQ_ACTIVE_INST_LDS") / m("SQ_WAVE_CYCLES")))
print(open("$OUT/pmc_attention_summary.txt").read())
PY
rm -rf $OUT/g1 $OUT/g2 $OUT/g3
