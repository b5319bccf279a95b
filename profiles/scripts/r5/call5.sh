set -x
mkdir -p gpurun_out/c5
export TANGO_TEST_THREADS=16
( timeout 900 python -m pytest tests/test_parity_batch_gpu.py -k "geglu_projection or (unet_and_loop and fp16)" tests/test_determinism_gpu.py tests/test_ops_gpu.py -k "geglu or stream or linear or unet_and_loop" -x -q -s -m gpu 2>&1 | grep -v Warning | tail -25 ) > gpurun_out/c5/tests_stream_spec.log 2>&1
tail -4 gpurun_out/c5/tests_stream_spec.log
timeout 400 python tools/profile_unet_ops.py --batch 32 --ab "TANGO_STREAM_SPEC=0;TANGO_STREAM_SPEC=1" --rounds 5 --grep "stream" --out gpurun_out/c5/stream_spec_ab_b32.txt > /dev/null 2>&1
timeout 400 python tools/profile_unet_ops.py --batch 8 --ab "TANGO_STREAM_SPEC=0;TANGO_STREAM_SPEC=1" --rounds 5 --grep "stream" --out gpurun_out/c5/stream_spec_ab_b8.txt > /dev/null 2>&1
head -8 gpurun_out/c5/stream_spec_ab_b32.txt | cut -c1-140; grep stream gpurun_out/c5/stream_spec_ab_b32.txt | cut -c1-140; grep -E "TOTAL|stream" gpurun_out/c5/stream_spec_ab_b8.txt | cut -c1-140
