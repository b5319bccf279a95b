#!/bin/bash
# round-3 GPU call 10: 4-wave (128x160 per wave) main-loop candidate vs the 8-wave ping-pong loop, bit-checked
O=gpurun_out/r3c10; mkdir -p $O
for shape in "262144 320 2560" "262144 320 11520" "65536 1280 1280" "16384 1280 2560" "65536 640 5120"; do
  timeout 120 ./build/loop_probe4w $shape 2>&1 | tee -a $O/probe4w.txt
done
