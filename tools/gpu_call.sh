#!/bin/bash
# gpurun call 27 of round 2: final numbers -- split-K test case, default bench (with cpu_baseline), bf16 and fp32 records
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2; mkdir -p $O
timeout 600 python -m pytest tests/test_determinism_gpu.py -m gpu -q -x -k "conv3x3_wide or wide_gemm_repeat" > $O/det_final.log 2>&1; echo "det rc=$?"; tail -2 $O/det_final.log
timeout 900 python bench.py > $O/bench_v26.json 2> $O/bench_v26.err; cat $O/bench_v26.json
timeout 600 python bench.py --dtype bf16 --denoise-steps 20 --no-cpu-baseline > $O/bench_v26_bf16.json 2> $O/bench_v26_bf16.err; cat $O/bench_v26_bf16.json | cut -c1-400
timeout 600 python bench.py --dtype fp32 --denoise-steps 10 --no-cpu-baseline > $O/bench_v26_fp32.json 2> $O/bench_v26_fp32.err; cat $O/bench_v26_fp32.json | cut -c1-400
