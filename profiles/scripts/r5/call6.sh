set -x
mkdir -p gpurun_out/c6
export TANGO_TEST_THREADS=16
( timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py tests/test_determinism_gpu.py tests/test_serving_gpu.py tests/test_text_encoder_gpu.py tests/test_stft_gpu.py tests/test_parity_batch_gpu.py -k "not unet_and_loop and not config5 and not text_buckets and not 100_steps" -x -q -m gpu 2>&1 | grep -v Warning | tail -8 ) > gpurun_out/c6/tests_epilogue_dispatch.log 2>&1
tail -3 gpurun_out/c6/tests_epilogue_dispatch.log
timeout 300 python bench.py --batch 1 --denoise-steps 100 --no-cpu-baseline --no-other-configs > gpurun_out/c6/bench_b1.json 2> gpurun_out/c6/bench_b1.err
timeout 300 python bench.py --batch 8 --denoise-steps 100 --no-cpu-baseline --no-other-configs > gpurun_out/c6/bench_b8.json 2> gpurun_out/c6/bench_b8.err
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c6/stats -o st -- python $R/bench.py --batch 32 --denoise-steps 4 --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > $R/gpurun_out/c6/stats.log 2>&1
DB=$(find $R/gpurun_out/c6/stats -name "*.db" | head -1)
python $R/tools/rocprof_summary.py "$DB" $R/gpurun_out/c6/kernel_stats.txt "python bench.py --batch 32 --denoise-steps 4 --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs"
rm -rf $R/gpurun_out/c6/stats
cd $R
python - <<'PY'
import json
for b in (1,8):
    d=json.load(open("gpurun_out/c6/bench_b%d.json"%b)); print("B=%d"%b, d["value"], d["ms_per_step"], d["roofline"]["kernel"].split(",")[-1])
PY
head -16 gpurun_out/c6/kernel_stats.txt | cut -c1-170
