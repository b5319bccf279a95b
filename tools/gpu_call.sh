#!/bin/bash
# gpurun call 4 of round 2: SIMD mapping probe, ping-pong with SIMD-based phases (A/B), T5 encoder parity, race DBG 4/5
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2; mkdir -p $O
timeout 60 build/simd_probe > $O/simd_probe.txt 2>&1; tail -2 $O/simd_probe.txt
timeout 600 python -m pytest tests/test_text_encoder_gpu.py tests/test_string_ckpt_gpu.py -m gpu -q -s > $O/t5.log 2>&1; echo "t5 rc=$?"; grep -E "rel err|passed|failed|Error" $O/t5.log | tail -12
TANGO_CONV_PP=1 timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "conv2d or linear" > $O/ops_pp2.log 2>&1; echo "ops(pp) rc=$?"; tail -1 $O/ops_pp2.log
for d in 4 5; do
  TANGO_STREAM_DBG=$d REPS=300 DTYPE=bf16 timeout 200 python tools/diag_stream_race.py > $O/race_dbg$d.txt 2>&1; echo "dbg=$d: $(tail -1 $O/race_dbg$d.txt)"
done
TANGO_CONV_PP=1 timeout 200 python tools/profile_unet_ops.py --out $O/unet_ops_pp_simd.txt > /dev/null 2>&1; head -1 $O/unet_ops_pp_simd.txt
TANGO_CONV_PP=0 TANGO_GEMM_PP=0 timeout 200 python tools/profile_unet_ops.py --out $O/unet_ops_nopp2.txt > /dev/null 2>&1; head -1 $O/unet_ops_nopp2.txt
