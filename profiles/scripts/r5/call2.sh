set -x
mkdir -p gpurun_out/c2
export TANGO_TEST_THREADS=16
( timeout 900 python -m pytest tests/test_gn_coop_gpu.py tests/test_attention_fp8_gpu.py tests/test_parallel_nccl_gpu.py -x -q -m gpu 2>&1 | grep -v Warning | tail -40 ) > gpurun_out/c2/tests_new.log 2>&1
tail -3 gpurun_out/c2/tests_new.log
# dual-chain denoise: parity (fp32 + fp16 at B = 32 / 8, graph == eager on the tiny config), then the A/B
( TANGO_UNET_CHAINS=2 timeout 1200 python -m pytest tests/test_parity_batch_gpu.py -k "unet_and_loop and (fp32 or fp16)" tests/test_engine_gpu.py -k "denoise or unet_and_loop" -x -q -s -m gpu 2>&1 | grep -v Warning | tail -40 ) > gpurun_out/c2/tests_dual.log 2>&1
tail -3 gpurun_out/c2/tests_dual.log
for b in 1 4 8 32; do
  n=100; [ $b = 32 ] && n=40
  for c in 1 2; do
    TANGO_UNET_CHAINS=$c timeout 300 python bench.py --batch $b --denoise-steps $n --no-cpu-baseline --no-other-configs > gpurun_out/c2/bench_b${b}_chains$c.json 2> gpurun_out/c2/bench_b${b}_chains$c.err
  done
done
TANGO_UNET_CHAINS=2 TANGO_FORCE_DMA_GEMM=1 timeout 300 python bench.py --batch 32 --denoise-steps 40 --no-cpu-baseline --no-other-configs > gpurun_out/c2/bench_b32_chains2_force.json 2> gpurun_out/c2/bench_b32_chains2_force.err
TANGO_UNET_CHAINS=2 TANGO_FORCE_DMA_GEMM=1 timeout 300 python bench.py --batch 8 --denoise-steps 100 --no-cpu-baseline --no-other-configs > gpurun_out/c2/bench_b8_chains2_force.json 2> gpurun_out/c2/bench_b8_chains2_force.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c2/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], "value %.3f launch-ms %s"%(d["value"], d["roofline"]["kernel"].split(",")[-1]))
    except Exception as e: print(f, "FAILED", e)
PY
tail -3 gpurun_out/c2/*.err | tail -30
