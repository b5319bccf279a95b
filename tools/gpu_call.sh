#!/bin/bash
# gpurun call 6 of round 2: persistent GEMM correctness + A/B, stream kernel with 32-row staging (1000 reps)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2; mkdir -p $O
TANGO_STRESS_REPS=20 timeout 900 python -m pytest tests/test_determinism_gpu.py -m gpu -q -x -k "linear_repeat or persistent or stream_linear" > $O/det_pers.log 2>&1; echo "det rc=$?"; tail -3 $O/det_pers.log
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "linear" > $O/ops_lin.log 2>&1; echo "ops rc=$?"; tail -1 $O/ops_lin.log
REPS=1000 DTYPE=bf16 timeout 300 python tools/diag_stream_race.py > $O/race_s32_bf16.txt 2>&1; echo "s32 bf16: $(tail -1 $O/race_s32_bf16.txt)"
REPS=500 DTYPE=fp16 timeout 300 python tools/diag_stream_race.py > $O/race_s32_fp16.txt 2>&1; echo "s32 fp16: $(tail -1 $O/race_s32_fp16.txt)"
TANGO_STREAM_NOFIX=1 REPS=300 DTYPE=bf16 timeout 300 python tools/diag_stream_race.py > $O/race_nofix2_bf16.txt 2>&1; echo "nofix bf16: $(tail -1 $O/race_nofix2_bf16.txt)"
timeout 200 python tools/profile_unet_ops.py --out $O/unet_ops_pers.txt > /dev/null 2>&1; head -1 $O/unet_ops_pers.txt
TANGO_NO_PERS_GEMM=1 timeout 200 python tools/profile_unet_ops.py --out $O/unet_ops_nopers.txt > /dev/null 2>&1; head -1 $O/unet_ops_nopers.txt
