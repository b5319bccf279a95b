#!/bin/bash
# gpurun call 11 of round 2: packed-fma GELU + dot2 LayerNorm statistics -- correctness, repeat-run stress, per-op table
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x > $O/ops_v24.log 2>&1; echo "ops rc=$?"; tail -2 $O/ops_v24.log
timeout 900 python -m pytest tests/test_determinism_gpu.py -m gpu -q -k "linear or gemm" > $O/det_v24.log 2>&1; echo "det rc=$?"; tail -2 $O/det_v24.log
REPS=500 DTYPE=bf16 timeout 300 python tools/diag_stream_race.py > $O/race_v24_bf16.txt 2>&1; echo "v24 bf16: $(tail -1 $O/race_v24_bf16.txt)"
REPS=500 DTYPE=fp16 timeout 300 python tools/diag_stream_race.py > $O/race_v24_fp16.txt 2>&1; echo "v24 fp16: $(tail -1 $O/race_v24_fp16.txt)"
timeout 200 python tools/profile_unet_ops.py --out $O/unet_ops_v24.txt > /dev/null 2>&1; head -1 $O/unet_ops_v24.txt
grep -E "N=2560 K=320|N=5120 K=640|N=10240 K=1280|ln\(stream\)" $O/unet_ops_v24.txt
timeout 900 python -m pytest tests/test_parity_full_gpu.py -m gpu -q -x -k "fp16 or bf16" > $O/parity_v24.log 2>&1; echo "parity rc=$?"; tail -3 $O/parity_v24.log
