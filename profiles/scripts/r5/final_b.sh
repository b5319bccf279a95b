#!/bin/bash
# the files of the GPU suite behind the one expectation that the first run of final.sh tripped over (fixed: tests/test_single_key_gpu.py)
set -x
export TANGO_TEST_THREADS=16
mkdir -p gpurun_out/final5b
( timeout 1200 python -m pytest tests/test_single_key_gpu.py tests/test_stft_gpu.py tests/test_string_ckpt_gpu.py tests/test_text_encoder_gpu.py tests/test_xattn_gpu.py -x -q -m gpu 2>&1 | grep -v Warning | tail -8 ) > gpurun_out/final5b/tests_rest.log 2>&1
tail -3 gpurun_out/final5b/tests_rest.log
