#!/bin/bash
# round-3 GPU call 1: the new parity / serving / front-end tests + same-box baseline per-op profiles (B = 32, 8, 1)
O=gpurun_out/r3c1; mkdir -p $O
export TANGO_STRESS_REPS=50
timeout 1300 python -m pytest tests/test_parity_batch_gpu.py tests/test_serving_gpu.py tests/test_text_encoder_gpu.py tests/test_stft_gpu.py \
  "tests/test_ops_gpu.py::test_groupnorm" -x -q -s --durations=25 > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
tail -60 $O/tests.log
for b in 32 8 1; do timeout 300 python tools/profile_unet_ops.py --batch $b --out $O/unet_ops_b$b.txt > /dev/null 2>&1; head -1 $O/unet_ops_b$b.txt; done
grep -i groupnorm $O/unet_ops_b8.txt | head
