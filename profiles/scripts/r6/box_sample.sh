# one more box of the pool: the default config-3 line of the final tree without the CPU leg / sub-records (200 steps, 1 warm-up + 1 timed pass)
mkdir -p gpurun_out/box
timeout 400 python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/box/bench_$1.json 2> gpurun_out/box/bench_$1.err
python -c "
import json; d=json.load(open('gpurun_out/box/bench_$1.json')); print(d['value'], d['roofline']['kernel'][-40:], d['roofline']['frac'])"
