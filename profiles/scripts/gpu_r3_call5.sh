#!/bin/bash
# round-3 GPU call 5: fused cross-attention block -- op parity, engine parity, A/B against the three-launch path
O=gpurun_out/r3c5; mkdir -p $O
timeout 900 python -m pytest tests/test_xattn_gpu.py tests/test_attention_fp8_gpu.py "tests/test_engine_gpu.py::test_unet_forward_large" \
  "tests/test_engine_gpu.py::test_unet_matches_reference_golden" "tests/test_ops_gpu.py::test_linear_layernorm_fused" -x -q -s --durations=8 > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
grep -E "passed|failed|rc=|rel err|rel diff|xattn|^E  " $O/tests.log | grep -v "print(" | tail -30
for b in 32 8 1; do
  timeout 200 python tools/profile_unet_ops.py --batch $b --out $O/ops_b${b}_fused.txt > /dev/null 2>&1
  TANGO_NO_XATTN_FUSED=1 timeout 200 python tools/profile_unet_ops.py --batch $b --out $O/ops_b${b}_unfused.txt > /dev/null 2>&1
  echo "== B=$b fused $(head -1 $O/ops_b${b}_fused.txt)"; grep -E "xattn" $O/ops_b${b}_fused.txt
  echo "== B=$b unfused $(head -1 $O/ops_b${b}_unfused.txt)"; grep -E "Skv=64 heads=5|linear\+ln\(wide\) M=[0-9]+ N=320 K=320|linear\+ln\(stream\) M=[0-9]+ N=320 K=320|linear\((wide|stream)\) M=[0-9]+ N=320 K=320|linear M=8192 N=320 K=320" $O/ops_b${b}_unfused.txt
done
