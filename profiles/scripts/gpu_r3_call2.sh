#!/bin/bash
# round-3 GPU call 2: Music UNet / fp8 attention / STFT tests; fp8 + XL bench lines; small-batch dispatch A/B (64x64 tiles, stream threshold)
O=gpurun_out/r3c2; mkdir -p $O
timeout 900 python -m pytest tests/test_music_unet_gpu.py tests/test_attention_fp8_gpu.py tests/test_stft_gpu.py \
  "tests/test_parity_full_gpu.py::test_config5_precision_bf16_with_fp8_attention" "tests/test_parity_full_gpu.py::test_config1_reduced_precision_ladder" \
  "tests/test_ops_gpu.py::test_linear" "tests/test_ops_gpu.py::test_linear_splitk" -q -s --durations=10 > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
grep -E "passed|failed|rc=|rel err|err vs|fp8 P.V|config 1|Music" $O/tests.log | tail -60
# ---- small-batch dispatch A/B (per-op profiles; same box) ----
for b in 1 8; do
  timeout 200 python tools/profile_unet_ops.py --batch $b --out $O/ops_b${b}_new.txt > /dev/null 2>&1
  TANGO_NO_SMALL_TILE=1 timeout 200 python tools/profile_unet_ops.py --batch $b --out $O/ops_b${b}_nosmall.txt > /dev/null 2>&1
  TANGO_STREAM_BIG_M=1 timeout 200 python tools/profile_unet_ops.py --batch $b --out $O/ops_b${b}_bigm.txt > /dev/null 2>&1
  head -1 $O/ops_b${b}_new.txt $O/ops_b${b}_nosmall.txt $O/ops_b${b}_bigm.txt | grep "^#"
done
# ---- bench lines: bf16 vs bf16 + fp8 attention (20 denoise steps), XL bf16 + fp8 at config 5's per-GPU shard ----
timeout 300 python bench.py --dtype bf16 --denoise-steps 20 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_bf16_b32_20.json 2>$O/bench1.err
timeout 300 python bench.py --dtype bf16 --fp8-attn --denoise-steps 20 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_bf16_fp8_b32_20.json 2>$O/bench2.err
timeout 300 python bench.py --xl --dtype bf16 --fp8-attn --batch 8 --denoise-steps 200 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_xl_bf16_fp8_b8_200.json 2>$O/bench3.err
for f in $O/bench_*.json; do python - <<PY
import json
try:
    d=json.loads(open("$f").read().strip().splitlines()[-1]); print("$f", round(d["value"],2), d["roofline"]["kernel"][-30:], round(d["roofline"]["frac"],4), d["config"]["workload"][:60])
except Exception as e: print("$f", "FAILED", e)
PY
done
