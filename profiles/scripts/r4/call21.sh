#!/bin/bash
# round-4 GPU call 21: bf16 (and bf16 + fp8 P.V) engines at B = 32 on the final tree, 20-step runs
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c21; mkdir -p $O
timeout 150 python bench.py --dtype bf16 --denoise-steps 20 --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > $O/bench_bf16_b32_20step.json 2>/dev/null
timeout 150 python bench.py --dtype bf16 --fp8-attn --denoise-steps 20 --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > $O/bench_bf16_fp8attn_b32_20step.json 2>/dev/null
for f in bench_bf16_b32_20step bench_bf16_fp8attn_b32_20step; do python - <<PY
import json
d = json.loads(open("$O/$f.json").read().strip().splitlines()[-1])
print("$f: %.1f TFLOP/s; %s" % (d["roofline"]["achieved"], d["roofline"]["kernel"].split(", ")[-1]))
PY
done
