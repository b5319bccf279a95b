set -x
mkdir -p gpurun_out/c1
export TANGO_TEST_THREADS=16
( timeout 900 python -m pytest tests/test_parity_batch_gpu.py -k "vae_and_vocoder" tests/test_gn_coop_gpu.py tests/test_attention_fp8_gpu.py tests/test_parallel_nccl_gpu.py -x -q -s -m gpu 2>&1 | grep -v Warning | tail -80 ) > gpurun_out/c1/tests.log 2>&1
# graph-steps A/B at B=1 (config 2) and B=8
for k in 1 10 50; do
  TANGO_GRAPH_STEPS=$k timeout 300 python bench.py --batch 1 --denoise-steps 100 --no-cpu-baseline --no-other-configs > gpurun_out/c1/bench_b1_k$k.json 2> gpurun_out/c1/bench_b1_k$k.err
done
for k in 1 10; do
  TANGO_GRAPH_STEPS=$k timeout 300 python bench.py --batch 8 --denoise-steps 200 --no-cpu-baseline --no-other-configs > gpurun_out/c1/bench_b8_k$k.json 2> gpurun_out/c1/bench_b8_k$k.err
done
# the driver's default line without the CPU leg (other_configs incl. the fp32 record)
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/c1/bench_default_nocpu.json 2> gpurun_out/c1/bench_default_nocpu.err
# per-op tables for planning
timeout 300 python tools/profile_unet_ops.py --batch 32 --out gpurun_out/c1/unet_ops_b32.txt > /dev/null 2>&1
timeout 300 python tools/profile_unet_ops.py --batch 1 --out gpurun_out/c1/unet_ops_b1.txt > /dev/null 2>&1
tail -5 gpurun_out/c1/tests.log
cat gpurun_out/c1/bench_b1_k*.json | cut -c1-400
