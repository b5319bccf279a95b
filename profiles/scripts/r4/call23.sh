#!/bin/bash
# round-4 GPU call 23: full-size UNet forward with outlier channels in the weights (fp32 / fp16 / bf16 engines vs the oracle)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c23; mkdir -p $O
timeout 280 python -m pytest tests/test_outlier_channels_gpu.py -q -m gpu -s > $O/outlier_channels.log 2>&1; echo "rc=$?"; grep -E "outlier-channel|passed|failed|assert" $O/outlier_channels.log
