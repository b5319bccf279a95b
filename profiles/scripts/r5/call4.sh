set -x
mkdir -p gpurun_out/c4
export TANGO_TEST_THREADS=16
( timeout 1500 python -m pytest tests/test_parity_batch_gpu.py -k "groupnorm_statistics or (unet_and_loop and (fp16 or bf16)) or config5" tests/test_determinism_gpu.py tests/test_duo_gpu.py -x -q -s -m gpu 2>&1 | grep -v Warning | tail -40 ) > gpurun_out/c4/tests_gnstats.log 2>&1
tail -4 gpurun_out/c4/tests_gnstats.log
for b in 32 8; do
  n=100; [ $b = 32 ] && n=40
  for sw in 1 0; do
    TANGO_NO_GN_PRODUCER_STATS=$sw timeout 300 python bench.py --batch $b --denoise-steps $n --no-cpu-baseline --no-other-configs > gpurun_out/c4/bench_b${b}_nogs$sw.json 2> gpurun_out/c4/bench_b${b}_nogs$sw.err
  done
done
timeout 300 python tools/profile_unet_ops.py --batch 32 --out gpurun_out/c4/unet_ops_b32_gnstats.txt > /dev/null 2>&1
timeout 400 python tools/profile_unet_ops.py --batch 32 --ab "TANGO_NO_GN_PRODUCER_STATS=1;TANGO_NO_GN_PRODUCER_STATS=0" --rounds 3 --out gpurun_out/c4/unet_ops_b32_gnstats_ab.txt > /dev/null 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c4/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], "value %.3f  %s"%(d["value"], d["roofline"]["kernel"].split(",")[-1]))
    except Exception as e: print(f, "FAILED", e)
PY
tail -3 gpurun_out/c4/*.err | tail -20
head -30 gpurun_out/c4/unet_ops_b32_gnstats_ab.txt | cut -c1-140
