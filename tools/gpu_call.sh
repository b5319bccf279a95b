#!/bin/bash
# one gpurun call of round 2: micro-benchmarks, new parity / determinism tests, per-op profile, short bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2; mkdir -p $O
timeout 120 build/mfma_war_repro 2000 512 > $O/mfma_war_repro.txt 2>&1; echo "repro rc=$?"
timeout 120 build/fill_bench > $O/fill_bench.txt 2>&1; echo "fill rc=$?"
timeout 1000 python -m pytest tests/test_parity_full_gpu.py tests/test_determinism_gpu.py tests/test_string_ckpt_gpu.py -m gpu -q -s --maxfail=25 > $O/newtests.log 2>&1; echo "tests rc=$?"
timeout 300 python tools/profile_unet_ops.py --out $O/unet_ops_v21.txt > /dev/null 2>&1; echo "prof rc=$?"
timeout 300 python bench.py --steps 1 --warmup 1 --denoise-steps 20 --no-cpu-baseline > $O/bench_v21.json 2> $O/bench_v21.err; echo "bench rc=$?"
tail -3 $O/mfma_war_repro.txt; tail -5 $O/newtests.log; head -3 $O/unet_ops_v21.txt; cat $O/bench_v21.json | cut -c1-400
