#!/bin/bash
# round-4 GPU call 8: duo stagger experiment, the measured default routing (duo where the 8-wave kernels run short of tiles), kernel-to-kernel gaps at B = 1 / 8 / 32
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c8; mkdir -p $O
timeout 300 python -m pytest tests/test_duo_gpu.py -q -m gpu -x > $O/tests_duo.log 2>&1; echo "duo tests rc=$?"; tail -2 $O/tests_duo.log
AB="TANGO_DUO_MAXK=0;TANGO_DUO_MAXK=640,TANGO_DUO_MASK=7;TANGO_DUO_MAXK=640,TANGO_DUO_MASK=7,TANGO_DUO_STAGGER=1;TANGO_DUO_MAXK=640,TANGO_DUO_MASK=7,TANGO_DUO_STAGGER=2;TANGO_DUO_MAXK=640,TANGO_DUO_MASK=7,TANGO_DUO_STAGGER=3;TANGO_DUO_MAXK=640,TANGO_DUO_MASK=7,TANGO_DUO_STAGGER=1,TANGO_DUO_DELAY_PCT=200;TANGO_DUO_MAXK=640,TANGO_DUO_MASK=7,TANGO_DUO_STAGGER=1,TANGO_DUO_DELAY_PCT=50"
timeout 600 python tools/profile_unet_ops.py --batch 32 --ab "$AB" --rounds 3 --grep "K=640|K=320" --out $O/duo_stagger_ab_b32.txt > /dev/null 2>$O/ab_st.err; echo "stagger ab rc=$?"; head -3 $O/duo_stagger_ab_b32.txt | cut -c1-260; grep -E "N=5120 K=640|N=640 K=640|N=320 K=320|N=960 K=320|N=1920 K=640" $O/duo_stagger_ab_b32.txt | cut -c1-260
for b in 32 8 1; do
  timeout 400 python tools/profile_unet_ops.py --batch $b --ab "TANGO_DUO_MAXK=0;TANGO_DUO_PRIO=0" --rounds 3 --grep "linear" --out $O/duo_default_ab_b$b.txt > /dev/null 2>$O/ab_def$b.err; echo "default ab b$b rc=$?"; head -3 $O/duo_default_ab_b$b.txt | cut -c1-120
done
timeout 600 python -m pytest tests/test_parity_batch_gpu.py -q -m gpu -x -s -k "benchmarked_batch and fp16" > $O/parity_batch_fp16.log 2>&1; echo "parity rc=$?"; grep -E "rel err|passed|failed" $O/parity_batch_fp16.log | tail -8
cd /tmp; export TMPDIR=/tmp
for cfg in "1 20" "8 6" "32 4"; do set -- $cfg
  CMD="python $R/bench.py --batch $1 --denoise-steps $2 --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs"
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/tr_b$1 -o st -- $CMD > $O/tr_b$1.log 2>&1
  DB=$(find $O/tr_b$1 -name "*.db" | head -1)
  python $R/tools/kernel_gaps.py "$DB" $O/kernel_gaps_b$1.txt "$CMD"; python $R/tools/rocprof_summary.py "$DB" $O/kernel_stats_b$1.txt "$CMD"
  rm -rf $O/tr_b$1; head -3 $O/kernel_gaps_b$1.txt | cut -c1-250
done
