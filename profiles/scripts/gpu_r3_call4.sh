#!/bin/bash
# round-3 GPU call 4: routing experiments for the level-0 K = 320 shapes on the 256 x 320 GEMM (B = 32 and B = 8) + fp8 flat-case diagnostic
O=gpurun_out/r3c4; mkdir -p $O
timeout 120 python tools/diag_fp8_flat.py > $O/fp8_flat_diag.txt 2>&1; cat $O/fp8_flat_diag.txt | tail -14
run() { name=$1; shift; env "$@" timeout 200 python tools/profile_unet_ops.py --batch $B --out $O/ops_b${B}_$name.txt > /dev/null 2>&1; echo "== B=$B $name $(head -1 $O/ops_b${B}_$name.txt)"; grep -E "N=320 K=320|N=960 K=320|N=2560 K=320" $O/ops_b${B}_$name.txt; }
B=32
run base X=1
run ln1 TANGO_EXP_WIDE_LN320=1
run ln3 TANGO_EXP_WIDE_LN320=3
B=8
run base X=1
run w256 TANGO_EXP_WIDE320_TILES=224
run w256ln3 TANGO_EXP_WIDE320_TILES=224 TANGO_EXP_WIDE_LN320=3
timeout 300 python -m pytest "tests/test_ops_gpu.py::test_linear_layernorm_fused" "tests/test_determinism_gpu.py::test_wide_gemm_ln_repeat" "tests/test_determinism_gpu.py::test_linear_qkv_vt_repeat" -q 2>&1 | tail -3
TANGO_EXP_WIDE_LN320=3 TANGO_EXP_WIDE320_TILES=1 TANGO_FORCE_DMA_GEMM=1 timeout 300 python -m pytest "tests/test_ops_gpu.py::test_linear_layernorm_fused" "tests/test_determinism_gpu.py::test_linear_qkv_vt_repeat" "tests/test_determinism_gpu.py::test_stream_linear_ln_repeat" -q 2>&1 | tail -3
