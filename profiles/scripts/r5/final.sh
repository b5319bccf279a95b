#!/bin/bash
# Round-5 end-of-round evidence on the final tree: full GPU suite, the driver's default bench line, rocprofv3 kernel stats of the bench
# command, HBM-side PMC passes (N = 2 and N = 6 steps), PMC view of the GEMM kernels incl. the level-0 streaming GEGLU projection.
set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/final7; mkdir -p $OUT
export TANGO_TEST_THREADS=16
cd $R
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -v Warning | tail -15 ) > $OUT/tests_gpu.log 2>&1
tail -3 $OUT/tests_gpu.log
timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python -c "
import json; d=json.load(open('$OUT/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'][-40:]); print({k:(round(v['value'],2), round(v['denoise_step_launch_ms'],2)) for k,v in d['other_configs'].items()}); print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
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
PMC_OUT=final7/pmc bash tools/pmc_op.sh run gemm_wide_pers_res_l1 gemm_wide_pers linear 65536 640 640 3 res > /dev/null 2>&1
PMC_OUT=final7/pmc2 bash tools/pmc_op.sh run stream_ln_geglu_l0 lin_stream linear_ln 262144 2560 320 3 geglu > /dev/null 2>&1
cat $OUT/pmc/summary.txt $OUT/pmc2/summary.txt | grep -E "##|derived|FETCH|WRITE"
