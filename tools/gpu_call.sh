#!/bin/bash
# gpurun call 5 of round 2: shipped stream-kernel fix (1000 reps), fast-erf GEGLU, profile + bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2; mkdir -p $O
REPS=1000 DTYPE=bf16 timeout 300 python tools/diag_stream_race.py > $O/race_fixed_bf16.txt 2>&1; echo "fixed bf16: $(tail -1 $O/race_fixed_bf16.txt)"
REPS=400 DTYPE=fp16 timeout 300 python tools/diag_stream_race.py > $O/race_fixed_fp16.txt 2>&1; echo "fixed fp16: $(tail -1 $O/race_fixed_fp16.txt)"
TANGO_STREAM_NOFIX=1 REPS=200 DTYPE=bf16 timeout 300 python tools/diag_stream_race.py > $O/race_nofix_bf16.txt 2>&1; echo "nofix bf16: $(tail -1 $O/race_nofix_bf16.txt)"
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "geglu or layernorm_fused" > $O/ops_geglu.log 2>&1; echo "geglu rc=$?"; tail -1 $O/ops_geglu.log
timeout 200 python tools/profile_unet_ops.py --out $O/unet_ops_v23.txt > /dev/null 2>&1; head -1 $O/unet_ops_v23.txt
timeout 300 python bench.py --steps 1 --warmup 1 --denoise-steps 20 --no-cpu-baseline > $O/bench_v23.json 2> $O/bench_v23.err; cut -c1-300 $O/bench_v23.json
timeout 300 python bench.py --steps 1 --warmup 1 --denoise-steps 50 --batch 1 --no-cpu-baseline > $O/bench_v23_b1.json 2> $O/bench_v23_b1.err; cut -c1-300 $O/bench_v23_b1.json
