#!/bin/bash
# round-4 final evidence on the frozen sources: per-op tables (B = 32 / 8 / 1) + per-op roofline, the default bench line (config 3 with
# cpu_baseline and other_configs), rocprofv3 kernel stats + HBM-side PMC passes of the bench command, smoke()
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4final; rm -rf $O; mkdir -p $O
cd $R
for b in 32 8 1; do timeout 200 python tools/profile_unet_ops.py --batch $b --out $O/unet_step_per_op_fp16_b$b.txt > /dev/null 2>&1; timeout 200 python tools/profile_unet_ops.py --batch $b --out $O/unet_step_per_op_fp16_b$b.txt > /dev/null 2>&1; head -1 $O/unet_step_per_op_fp16_b$b.txt; done
python tools/per_op_roofline.py $O/unet_step_per_op_fp16_b32.txt --batch2 64 --out $O/per_op_roofline_b32.txt > /dev/null 2>&1; tail -3 $O/per_op_roofline_b32.txt | cut -c1-160
timeout 1500 python bench.py > $O/bench_b32_200step.json 2> $O/bench_b32_200step.err; echo "bench rc=$?"; tail -c 1800 $O/bench_b32_200step.json
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
sed -i 's#gpurun_out/final[0-9a-z]*#gpurun_out/r4final#' tools/final_profiles.sh
bash tools/final_profiles.sh
python - <<PY
import json, re, sys
sys.path.insert(0, "$R")
import bench
tot = {}
for line in open("$O/pmc_totals.txt"):
    m = re.match(r"N=(\d+) (\w+) sum ([\d.]+)", line)
    if m: tot[(int(m.group(1)), m.group(2))] = float(m.group(3))
try:
    f = (tot[(6, "FETCH_SIZE")] - tot[(2, "FETCH_SIZE")]) / 4 * 1024
    w = (tot[(6, "WRITE_SIZE")] - tot[(2, "WRITE_SIZE")]) / 4 * 1024
    rec = {"batch": 32, "dtype": "fp16", "xl": False, "fp8_attn": False, "src_sha16": bench.kernel_source_sha16(),
           "fetch_raw_bytes": f, "write_bytes": w, "bytes_per_step": 2 * f + w, "bytes_per_step_raw": f + w}
    json.dump(rec, open("$O/hbm_traffic_record.json", "w"), indent=1)
    print("HBM-side bytes per denoise step: raw %.1f GB, corrected %.1f GB (sources %s)" % ((f + w) / 1e9, (2 * f + w) / 1e9, rec["src_sha16"]))
except KeyError as e:
    print("PMC totals incomplete", e, tot)
PY
