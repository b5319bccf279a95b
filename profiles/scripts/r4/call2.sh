#!/bin/bash
# round-4 GPU call 2: (a) redefined schedules 2 (DMAs at the very start of the read part) and 3 (paired DMAs) under the repeat-run
# tests, (b) A/B 0 / 1 / 2 / 3 at B = 32 and 0 / 1 at B = 8, (c) short bench run on the new default (1), (d) PMC view of the level-0
# and a long-K wide conv on schedule 1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c2; mkdir -p $O
for S in 2 3; do
  TANGO_WIDE_SCHED=$S timeout 400 python -m pytest tests/test_determinism_gpu.py -m gpu -q -x -k "wide_gemm or conv3x3_wide" > $O/det_sched$S.log 2>&1
  echo "sched $S determinism rc=$?"; tail -2 $O/det_sched$S.log; grep -E "rel err|differs" $O/det_sched$S.log | head -3
done
timeout 500 python tools/profile_unet_ops.py --ab "TANGO_WIDE_SCHED=0;TANGO_WIDE_SCHED=1;TANGO_WIDE_SCHED=2;TANGO_WIDE_SCHED=3" --rounds 3 \
  --grep "conv3x3|linear" --out $O/ab_sched_b32.txt > /dev/null 2> $O/ab_err.log; echo "ab rc=$?"; head -12 $O/ab_sched_b32.txt
timeout 300 python tools/profile_unet_ops.py --batch 8 --ab "TANGO_WIDE_SCHED=0;TANGO_WIDE_SCHED=1" --rounds 3 \
  --grep "conv3x3|linear" --out $O/ab_sched_b8.txt > /dev/null 2>> $O/ab_err.log; echo "ab8 rc=$?"; head -8 $O/ab_sched_b8.txt
timeout 400 python bench.py --denoise-steps 20 --no-cpu-baseline --no-other-configs > $O/bench_b32_20step.json 2> $O/bench_err.log; echo "bench rc=$?"; cut -c1-400 $O/bench_b32_20step.json
PMC_OUT=r4c2/pmc_l0 bash tools/pmc_op.sh run conv_wide_l0_sch1 conv3x3_wide conv 64 320 256 16 320 3 > $O/pmc_l0.log 2>&1; tail -3 $O/pmc_l0.log
PMC_OUT=r4c2/pmc_l1 bash tools/pmc_op.sh run conv_wide_l1_sch1 conv3x3_wide conv 64 640 128 8 640 3 > $O/pmc_l1.log 2>&1; tail -3 $O/pmc_l1.log
