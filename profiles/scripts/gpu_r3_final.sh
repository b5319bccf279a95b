#!/bin/bash
# round-3 final evidence run (one gpurun call): the driver's GPU test command, smoke, the default bench line, the other BASELINE
# configs with the same binary, the per-op table, then rocprofv3 kernel stats + HBM-side PMC passes (tools/final_profiles.sh)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=15 > $O/tests_gpu.log 2>&1; echo "pytest rc=$?" >> $O/tests_gpu.log
tail -4 $O/tests_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_b32_200step.json 2> $O/bench_b32_200step.err; tail -c 1800 $O/bench_b32_200step.json
timeout 300 python bench.py --batch 1 --denoise-steps 100 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_b1_100step.json 2>/dev/null
timeout 300 python bench.py --batch 8 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_b8_200step.json 2>/dev/null
timeout 300 python bench.py --xl --dtype bf16 --fp8-attn --batch 8 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_config5_shard_xl_bf16_fp8attn_b8_200step.json 2>/dev/null
timeout 300 python bench.py --dtype bf16 --denoise-steps 20 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_bf16_b32_20step.json 2>/dev/null
for f in bench_b1_100step bench_b8_200step bench_config5_shard_xl_bf16_fp8attn_b8_200step bench_bf16_b32_20step; do python - <<PY
import json
try:
    d = json.loads(open("$O/$f.json").read().strip().splitlines()[-1])
    print("$f: %.2f %s, roofline %.1f TFLOP/s; %s" % (d["value"], d["unit"], d["roofline"]["achieved"], d["roofline"]["kernel"].split(", ")[-1]))
except Exception as e:
    print("$f: FAILED", e)
PY
done
for b in 32 8 1; do timeout 200 python tools/profile_unet_ops.py --batch $b --out $O/unet_step_per_op_fp16_b$b.txt > /dev/null 2>&1; head -1 $O/unet_step_per_op_fp16_b$b.txt; done
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
    f = (tot[(6, "FETCH_SIZE")] - tot[(2, "FETCH_SIZE")]) / 4 * 1024      # KiB units
    w = (tot[(6, "WRITE_SIZE")] - tot[(2, "WRITE_SIZE")]) / 4 * 1024
    rec = {"batch": 32, "dtype": "fp16", "xl": False, "fp8_attn": False, "src_sha16": bench.kernel_source_sha16(),
           "fetch_raw_bytes": f, "write_bytes": w, "bytes_per_step": 2 * f + w, "bytes_per_step_raw": f + w}
    json.dump(rec, open("$O/hbm_traffic_record.json", "w"), indent=1)
    print("HBM-side bytes per denoise step: raw %.1f GB, with the gfx950 FETCH_SIZE x2 correction %.1f GB (sources %s)" % ((f + w) / 1e9, (2 * f + w) / 1e9, rec["src_sha16"]))
except KeyError as e:
    print("PMC totals incomplete", e, tot)
PY
