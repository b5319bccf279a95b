#!/bin/bash
# round-4 GPU call 17: attention softmax row sums on the matrix pipe (MSUM) -- op tests with it forced, A/B at B = 32 / 8 / 1, parity
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c17; mkdir -p $O
timeout 300 env TANGO_ATTN_MSUM=1 python -m pytest tests/test_ops_gpu.py tests/test_determinism_gpu.py -q -m gpu -x -s -k "attention" > $O/tests_attn_msum.log 2>&1; echo "attn tests (MSUM) rc=$?"; tail -2 $O/tests_attn_msum.log
for b in 32 8 1; do
  timeout 300 python tools/profile_unet_ops.py --batch $b --ab "TANGO_ATTN_MSUM=0;TANGO_ATTN_MSUM=1" --rounds 5 --grep "attention" --out $O/attn_msum_ab_b$b.txt > /dev/null 2>$O/ab$b.err; echo "ab b$b rc=$?"; head -3 $O/attn_msum_ab_b$b.txt | cut -c1-110; grep -E "^attention|family attention" $O/attn_msum_ab_b$b.txt | cut -c1-110
done
timeout 600 env TANGO_ATTN_MSUM=1 python -m pytest tests/test_parity_batch_gpu.py -q -m gpu -x -s -k "benchmarked_batch and (fp16 or bf16)" > $O/parity_batch_msum.log 2>&1; echo "parity rc=$?"; grep -E "rel err|passed|failed" $O/parity_batch_msum.log | tail -14
