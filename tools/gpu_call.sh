#!/bin/bash
# gpurun call 2 of round 2: correctness of the ping-pong loops, race characterisation, A/B per-op profiles
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "conv2d or linear" > $O/ops_pp.log 2>&1; echo "ops rc=$?"; tail -2 $O/ops_pp.log
TANGO_STRESS_REPS=30 timeout 600 python -m pytest tests/test_determinism_gpu.py -m gpu -q -k "conv3x3 or linear_repeat or large_mean" > $O/det_pp.log 2>&1; echo "det rc=$?"; tail -4 $O/det_pp.log
timeout 300 python -m pytest tests/test_string_ckpt_gpu.py "tests/test_parity_full_gpu.py::test_denoise_shard_invariance_on_device_noise" -m gpu -q -s > $O/str.log 2>&1; echo "str rc=$?"; tail -4 $O/str.log
REPS=300 DTYPE=bf16 timeout 200 python tools/diag_stream_race.py > $O/race_bf16.txt 2>&1; tail -3 $O/race_bf16.txt
REPS=300 DTYPE=bf16 TANGO_NO_STAGED_EPILOGUE=1 timeout 200 python tools/diag_stream_race.py > $O/race_bf16_nostage.txt 2>&1; tail -2 $O/race_bf16_nostage.txt
REPS=200 DTYPE=fp16 timeout 200 python tools/diag_stream_race.py > $O/race_fp16.txt 2>&1; tail -2 $O/race_fp16.txt
timeout 200 python tools/profile_unet_ops.py --out $O/unet_ops_pp.txt > /dev/null 2>&1; echo "prof rc=$?"; head -1 $O/unet_ops_pp.txt
TANGO_CONV_PP=0 TANGO_GEMM_PP=0 timeout 200 python tools/profile_unet_ops.py --out $O/unet_ops_nopp.txt > /dev/null 2>&1; head -1 $O/unet_ops_nopp.txt
