#!/bin/bash
# Round-6 end-of-round evidence on the final tree, one gpurun call: smoke, the default bench line, rocprofv3 kernel stats + HBM-side PMC passes
# (tools/final_profiles.sh), per-op tables B = 32 / 8 / 1 with the per-op roofline, VAE / vocoder per-op tables, PMC of the activation-stationary kernels.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${FINAL_DIR:-final2}; mkdir -p $OUT; cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 600 $OUT/bench.json
for b in 32 8 1; do python tools/profile_unet_ops.py --batch $b --out $OUT/unet_step_per_op_fp16_b$b.txt > /dev/null 2>&1; done
python tools/per_op_roofline.py $OUT/unet_step_per_op_fp16_b32.txt --out $OUT/per_op_roofline_b32.txt | tail -12
python tools/profile_vae_vocoder_ops.py --batch 32 > $OUT/vae_vocoder_per_op_b32.txt 2>&1
bash tools/final_profiles.sh > $OUT/final_profiles.log 2>&1; cp gpurun_out/final/kernel_stats.txt $OUT/; cp gpurun_out/final/pmc_totals.txt $OUT/; cat $OUT/pmc_totals.txt
bash tools/r6_pmc_stat_kernels.sh > $OUT/pmc_stat.log 2>&1; cp gpurun_out/r6_pmc_stat/summary.txt $OUT/pmc_stat_kernels_summary.txt; cat $OUT/pmc_stat_kernels_summary.txt
