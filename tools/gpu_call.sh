#!/bin/bash
# gpurun call 7 of round 2: race hunt -- asm-FMA build of the LN epilogue, plain variant, K = 320 variant
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2; mkdir -p $O
REPS=1000 DTYPE=bf16 timeout 300 python tools/diag_stream_race.py > $O/race_asmfma_bf16.txt 2>&1; echo "asm-fma LN bf16 K=640: $(tail -1 $O/race_asmfma_bf16.txt)"
TANGO_STREAM_NOFIX=1 REPS=300 DTYPE=bf16 timeout 300 python tools/diag_stream_race.py > $O/race_nofix3_bf16.txt 2>&1; echo "nofix LN bf16 K=640: $(tail -1 $O/race_nofix3_bf16.txt)"
REPS=1000 DTYPE=fp16 timeout 300 python tools/diag_stream_race.py > $O/race_asmfma_fp16.txt 2>&1; echo "asm-fma LN fp16 K=640: $(tail -1 $O/race_asmfma_fp16.txt)"
PLAIN=1 REPS=1000 DTYPE=bf16 timeout 300 python tools/diag_stream_race.py > $O/race_plain_bf16.txt 2>&1; echo "plain bf16 K=640: $(tail -1 $O/race_plain_bf16.txt)"
SHAPE=5000,1920,320 REPS=1000 DTYPE=bf16 timeout 300 python tools/diag_stream_race.py > $O/race_k320_bf16.txt 2>&1; echo "asm-fma LN bf16 K=320: $(tail -1 $O/race_k320_bf16.txt)"
SHAPE=5000,1920,320 PLAIN=1 REPS=600 DTYPE=fp16 timeout 300 python tools/diag_stream_race.py > $O/race_k320_plain_fp16.txt 2>&1; echo "plain fp16 K=320: $(tail -1 $O/race_k320_plain_fp16.txt)"
