#!/bin/bash
# round-3 GPU call 3: re-run of the fixed tests (Music UNet, fp8 attention, DDIM eta); shipped small-batch dispatch; level-0 plain-linear kernel A/B at B = 32
O=gpurun_out/r3c3; mkdir -p $O
timeout 900 python -m pytest tests/test_music_unet_gpu.py tests/test_attention_fp8_gpu.py tests/test_stft_gpu.py \
  "tests/test_engine_gpu.py::test_denoise_loop_ddim_eta" "tests/test_engine_gpu.py::test_denoise_loop_tiny" "tests/test_ops_gpu.py::test_linear" \
  "tests/test_determinism_gpu.py::test_linear_repeat" -q -s --durations=8 > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
grep -E "passed|failed|rc=|rel err|err vs|fp8 P.V|off-grid|DDIM|Music|^E  " $O/tests.log | grep -v "print(" | tail -50
for b in 1 8; do timeout 200 python tools/profile_unet_ops.py --batch $b --out $O/ops_b${b}.txt > /dev/null 2>&1; head -1 $O/ops_b${b}.txt; done
timeout 200 python tools/profile_unet_ops.py --batch 32 --out $O/ops_b32_base.txt > /dev/null 2>&1
TANGO_NO_STREAM=1 timeout 200 python tools/profile_unet_ops.py --batch 32 --out $O/ops_b32_nostream.txt > /dev/null 2>&1
TANGO_NO_STREAM=1 TANGO_NO_WIDE_GEMM=1 timeout 200 python tools/profile_unet_ops.py --batch 32 --out $O/ops_b32_dma.txt > /dev/null 2>&1
TANGO_NO_STREAM=1 TANGO_NO_WIDE_GEMM=1 TANGO_NO_DMA_GEMM=1 timeout 200 python tools/profile_unet_ops.py --batch 32 --out $O/ops_b32_generic.txt > /dev/null 2>&1
for f in base nostream dma generic; do echo "== $f $(head -1 $O/ops_b32_$f.txt)"; grep -E "M=262144 N=320 K=320|M=262144 N=320 K=640|M=65536 N=640 K=640 " $O/ops_b32_$f.txt; done
