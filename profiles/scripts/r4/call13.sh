#!/bin/bash
# round-4 GPU call 13: GEGLU projections of levels 1-2 with external LayerNorm statistics (ln_stats + gemm_wide XS) -- tests, A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c13; mkdir -p $O
timeout 900 python -m pytest tests/test_determinism_gpu.py tests/test_single_key_gpu.py -q -m gpu -x -s -k "xstats or to_out_epilogue" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|rel diff|rel err" $O/tests.log | tail -6
for b in 32 8; do
  timeout 300 python tools/profile_unet_ops.py --batch $b --ab "TANGO_NO_LN_XSTATS=1;TANGO_NO_LN_XSTATS=0" --rounds 5 --grep "N=5120|N=10240|layernorm|ln_stats" --out $O/xstats_ab_b$b.txt > /dev/null 2>$O/ab$b.err; echo "ab b$b rc=$?"; head -3 $O/xstats_ab_b$b.txt | cut -c1-110; grep -E "^layernorm|^ln_stats|^linear|family layernorm|family ln_stats" $O/xstats_ab_b$b.txt | cut -c1-110 | head -14
done
timeout 600 python -m pytest tests/test_parity_batch_gpu.py -q -m gpu -x -s -k "benchmarked_batch and (fp16 or bf16)" > $O/parity_batch.log 2>&1; echo "parity rc=$?"; grep -E "passed|failed" $O/parity_batch.log | tail -3
