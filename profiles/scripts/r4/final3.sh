#!/bin/bash
# round-4 last call: the whole GPU suite on the final tree, then the HBM-side PMC passes of the bench command keyed to its source hash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4final3; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/ -q -m gpu > $O/tests_gpu.log 2>&1; echo "suite rc=$?"
grep -E "passed|failed" $O/tests_gpu.log | tail -2; grep -E "^FAILED|^ERROR" $O/tests_gpu.log | head -20
cd /tmp; export TMPDIR=/tmp
for n in 2 6; do for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_${c}_$n -o p --output-format csv -- python $R/bench.py --batch 32 --denoise-steps $n --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > $O/pmc_${c}_$n.log 2>&1
  python - <<PY
import csv,glob
f=glob.glob("$O/pmc_${c}_$n/*counter_collection.csv")
tot=0.0; nd=0
for r in csv.DictReader(open(f[0])):
    if r['Counter_Name']=="$c": tot+=float(r['Counter_Value']); nd+=1
open("$O/pmc_totals.txt","a").write("N=$n $c sum %.1f dispatches %d\n"%(tot,nd))
PY
  rm -rf $O/pmc_${c}_$n
done; done
cat $O/pmc_totals.txt
cd $R
python - <<PY
import json, re, sys
sys.path.insert(0, "$R")
import bench
tot = {}
for line in open("$O/pmc_totals.txt"):
    m = re.match(r"N=(\d+) (\w+) sum ([\d.]+)", line)
    if m: tot[(int(m.group(1)), m.group(2))] = float(m.group(3))
f = (tot[(6, "FETCH_SIZE")] - tot[(2, "FETCH_SIZE")]) / 4 * 1024
w = (tot[(6, "WRITE_SIZE")] - tot[(2, "WRITE_SIZE")]) / 4 * 1024
rec = {"batch": 32, "dtype": "fp16", "xl": False, "fp8_attn": False, "src_sha16": bench.kernel_source_sha16(),
       "fetch_raw_bytes": f, "write_bytes": w, "bytes_per_step": 2 * f + w, "bytes_per_step_raw": f + w}
json.dump(rec, open("$O/hbm_traffic_record.json", "w"), indent=1)
print("HBM-side bytes per denoise step: raw %.1f GB, corrected %.1f GB (sources %s)" % ((f + w) / 1e9, (2 * f + w) / 1e9, rec["src_sha16"]))
PY
