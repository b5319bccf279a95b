#!/bin/bash
# Round-end evidence: rocprofv3 kernel stats of the bench command (4 denoise steps) + HBM-side PMC passes (N=2 and N=6 steps).
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/final; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --batch 32 --denoise-steps 4 --steps 1 --warmup 1 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o st -- $CMD > $OUT/stats.log 2>&1
DB=$(find $OUT/stats -name "*.db" | head -1)
python $R/tools/rocprof_summary.py "$DB" $OUT/kernel_stats.txt "python bench.py --batch 32 --denoise-steps 4 --steps 1 --warmup 1 --no-cpu-baseline"
rm -rf $OUT/stats
for n in 2 6; do for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_${c}_$n -o p --output-format csv -- python $R/bench.py --batch 32 --denoise-steps $n --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_${c}_$n.log 2>&1
  python - <<PY
import csv,glob
f=glob.glob("$OUT/pmc_${c}_$n/*counter_collection.csv")
tot=0.0; nd=0
for r in csv.DictReader(open(f[0])):
    if r['Counter_Name']=="$c": tot+=float(r['Counter_Value']); nd+=1
open("$OUT/pmc_totals.txt","a").write("N=$n $c sum %.1f dispatches %d\n"%(tot,nd))
PY
  rm -rf $OUT/pmc_${c}_$n
done; done
cat $OUT/pmc_totals.txt; head -12 $OUT/kernel_stats.txt | cut -c1-180
