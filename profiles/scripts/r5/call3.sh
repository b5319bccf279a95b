set -x
mkdir -p gpurun_out/c3
export TANGO_TEST_THREADS=16
# CFG-shared prefix (default on): parity through denoise at B = 32 / 8 / 1 and the tiny-config loops
( timeout 1500 python -m pytest tests/test_parity_batch_gpu.py -k "unet_and_loop or config5" tests/test_engine_gpu.py -k "denoise" tests/test_single_key_gpu.py tests/test_parity_full_gpu.py -k "config1 or config_1 or full" -x -q -s -m gpu 2>&1 | grep -v Warning | tail -60 ) > gpurun_out/c3/tests_shared.log 2>&1
tail -4 gpurun_out/c3/tests_shared.log
for b in 32 8 1; do
  n=100; [ $b = 32 ] && n=40
  for sw in 1 0; do
    TANGO_NO_CFG_SHARED=$sw timeout 300 python bench.py --batch $b --denoise-steps $n --no-cpu-baseline --no-other-configs > gpurun_out/c3/bench_b${b}_noshared$sw.json 2> gpurun_out/c3/bench_b${b}_noshared$sw.err
  done
done
timeout 300 python tools/profile_unet_ops.py --batch 32 --out gpurun_out/c3/unet_ops_b32_shared.txt > /dev/null 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c3/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], "value %.3f  %s  executed %.0f GF"%(d["value"], d["roofline"]["kernel"].split(",")[-1], d["roofline"]["executed_gflop"]))
    except Exception as e: print(f, "FAILED", e)
PY
tail -3 gpurun_out/c3/*.err | tail -20
