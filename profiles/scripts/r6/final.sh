#!/bin/bash
# Round-6 end-of-round evidence on the final tree (ONE gpurun call): the driver's default bench line, smoke(), rocprofv3 kernel stats of the
# bench command, HBM-side PMC passes (N = 2 and N = 6 steps), per-op tables of the UNet step (B = 32 / 8 / 1) and of the VAE decoder /
# HiFi-GAN (B = 32 / 1), per-op roofline, PMC view of the wide conv (pipelined default) and the persistent GEMM.
set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-final}; mkdir -p $OUT
export TANGO_TEST_THREADS=16
cd $R
timeout 1800 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python -c "
import json; d=json.load(open('$OUT/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'][-40:]); print({k:(round(v['value'],2), round(v['denoise_step_launch_ms'],2)) for k,v in d['other_configs'].items()}); print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['roofline']['same_precision']['value'], d['text_encoder_ms'])"
python -c "from __graft_entry__ import smoke; smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --batch 32 --denoise-steps 4 --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o st -- $CMD > $OUT/stats.log 2>&1
DB=$(find $OUT/stats -name "*.db" | head -1)
python $R/tools/rocprof_summary.py "$DB" $OUT/kernel_stats.txt "python bench.py --batch 32 --denoise-steps 4 --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs"
rm -rf $OUT/stats
rm -f $OUT/pmc_totals.txt
for n in 2 6; do for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_${c}_$n -o p --output-format csv -- python $R/bench.py --batch 32 --denoise-steps $n --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > $OUT/pmc_${c}_$n.log 2>&1
  python - <<PY
import csv,glob
f=glob.glob("$OUT/pmc_${c}_$n/**/*counter_collection.csv", recursive=True)
tot=0.0; nd=0
for r in csv.DictReader(open(f[0])):
    if r['Counter_Name']=="$c": tot+=float(r['Counter_Value']); nd+=1
open("$OUT/pmc_totals.txt","a").write("N=$n $c sum %.1f dispatches %d\n"%(tot,nd))
PY
  rm -rf $OUT/pmc_${c}_$n
done; done
cat $OUT/pmc_totals.txt; head -14 $OUT/kernel_stats.txt | cut -c1-180
cd $R
timeout 300 python tools/profile_unet_ops.py --batch 32 --out $OUT/unet_ops_b32.txt > /dev/null 2>&1
timeout 300 python tools/profile_unet_ops.py --batch 8 --out $OUT/unet_ops_b8.txt > /dev/null 2>&1
timeout 300 python tools/profile_unet_ops.py --batch 1 --out $OUT/unet_ops_b1.txt > /dev/null 2>&1
timeout 300 python tools/per_op_roofline.py $OUT/unet_ops_b32.txt > $OUT/per_op_roofline_b32.txt 2>&1
timeout 300 python tools/profile_vae_vocoder_ops.py --batch 32 --out $OUT/vae_vocoder_per_op_b32.txt > /dev/null 2>&1
timeout 300 python tools/profile_vae_vocoder_ops.py --batch 1 --out $OUT/vae_vocoder_per_op_b1.txt > /dev/null 2>&1
PMC_OUT=${1:-final}/pmc bash tools/pmc_op.sh run conv_wide_l0 conv3x3_wide conv 64 320 256 16 320 3 > /dev/null 2>&1
PMC_OUT=${1:-final}/pmc2 bash tools/pmc_op.sh run conv_wide_l1 conv3x3_wide conv 64 640 128 8 640 3 > /dev/null 2>&1
PMC_OUT=${1:-final}/pmc3 bash tools/pmc_op.sh run gemm_wide_pers_res_l1 gemm_wide_pers linear 65536 640 640 3 res > /dev/null 2>&1
cat $OUT/pmc/summary.txt $OUT/pmc2/summary.txt $OUT/pmc3/summary.txt 2>/dev/null | grep -E "##|derived|FETCH|WRITE"
head -3 $OUT/unet_ops_b32.txt; grep "^#" $OUT/vae_vocoder_per_op_b*.txt
