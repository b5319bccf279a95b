#!/bin/bash
# round-3 GPU call 13: B = 8 dispatch thresholds (config 5's per-GPU shard): split-K cut-off, minimum tile counts of the 8-wave GEMMs
O=gpurun_out/r3c13; mkdir -p $O
ROWS="M=4096 N=1280 K=1280|M=4096 N=3840 K=1280|M=16384 N=640 K=640|M=4096 N=1280 K=5120|M=4096 N=1280 K=2560|M=16384 N=640 K=2560|M=1024 N=1280"
run() { label=$1; shift; env "$@" timeout 200 python tools/profile_unet_ops.py --batch 8 --out $O/ops_b8_$label.txt > /dev/null 2>&1
  echo "== $label: $(head -1 $O/ops_b8_$label.txt)"; grep -E "$ROWS" $O/ops_b8_$label.txt | grep -v conv; }
run base A=1
run wide192 TANGO_EXP_WIDE_MINTILES=192
run wide128 TANGO_EXP_WIDE_MINTILES=128
run nosplit256 TANGO_EXP_NOSPLIT_TILES=256
run nosplit128 TANGO_EXP_NOSPLIT_TILES=128
run dma256 TANGO_EXP_DMA_MINTILES=256 TANGO_NO_STREAM=1
run base2 A=1
