#!/bin/bash
# round-3 final evidence, second pass (after the B = 8 dispatch thresholds of call 13): parity at the benchmarked batches on the new
# routing, per-op tables, the default bench line, config 5's shard, rocprofv3 stats + HBM-side PMC passes on THIS tree
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final2; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_parity_batch_gpu.py tests/test_determinism_gpu.py -k "(benchmarked_batch and fp16) or test_linear_repeat" -x -q -s > $O/tests_routing.log 2>&1; echo "pytest rc=$?" >> $O/tests_routing.log
grep -E "passed|failed|rc=|prompt|^E  " $O/tests_routing.log | grep -v "print(" | tail -16
for b in 8 32 1; do timeout 200 python tools/profile_unet_ops.py --batch $b --out $O/unet_step_per_op_fp16_b$b.txt > /dev/null 2>&1; timeout 200 python tools/profile_unet_ops.py --batch $b --out $O/unet_step_per_op_fp16_b$b.txt > /dev/null 2>&1; head -1 $O/unet_step_per_op_fp16_b$b.txt; done
timeout 900 python bench.py > $O/bench_b32_200step.json 2> $O/bench_b32_200step.err; tail -c 900 $O/bench_b32_200step.json
timeout 300 python bench.py --batch 8 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_b8_200step.json 2>/dev/null
timeout 300 python bench.py --xl --dtype bf16 --fp8-attn --batch 8 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_config5_shard_xl_bf16_fp8attn_b8_200step.json 2>/dev/null
for f in bench_b8_200step bench_config5_shard_xl_bf16_fp8attn_b8_200step; do python - <<PY
import json
try:
    d = json.loads(open("$O/$f.json").read().strip().splitlines()[-1])
    print("$f: %.2f %s, roofline %.1f TFLOP/s; %s" % (d["value"], d["unit"], d["roofline"]["achieved"], d["roofline"]["kernel"].split(", ")[-1]))
except Exception as e:
    print("$f: FAILED", e)
PY
done
sed -i 's#gpurun_out/final#gpurun_out/final2#' tools/final_profiles.sh
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
