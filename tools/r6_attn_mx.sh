#!/bin/bash
# Round 6, row g1: the S = 4096 self-attention site (B2 = 64 at config 3 -> here B = 16 x 5 heads, same kernel, quarter the grid... full: 64)
# with P.V in the engine dtype / on the non-scaled fp8 MFMA / on the MX instruction (two forms); kernel durations from rocprofv3.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-attn_mx}; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
run() {  # name, env..., -- args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/$name -o st -- python $R/tools/bench_ops.py "$@" > $OUT/$name.log 2>&1
  DB=$(find $OUT/$name -name "*.db" | head -1)
  python $R/tools/rocprof_summary.py "$DB" $OUT/$name.txt "${envs[*]} bench_ops.py $*" > /dev/null 2>&1
  rm -rf $OUT/$name
  echo "== $name"; grep -E "attn_kernel" $OUT/$name.txt | cut -c1-200 | head -3
}
for dt in bf16 fp16; do
  run s4096_${dt}_plain BENCH_DTYPE=$dt -- attention 64 5 4096 8 0
  run s4096_${dt}_fp8 BENCH_DTYPE=$dt -- attention 64 5 4096 8 1
  run s4096_${dt}_mx_qb2 BENCH_DTYPE=$dt TANGO_ATTN_X8_QB=2 -- attention 64 5 4096 8 3
  run s4096_${dt}_mx_qb1 BENCH_DTYPE=$dt TANGO_ATTN_X8_QB=1 -- attention 64 5 4096 8 3
done
run s1024_bf16_plain BENCH_DTYPE=bf16 -- attention 64 10 1024 8 0
run s1024_bf16_fp8 BENCH_DTYPE=bf16 -- attention 64 10 1024 8 1
run s1024_bf16_mx_qb2 BENCH_DTYPE=bf16 TANGO_ATTN_X8_QB=2 -- attention 64 10 1024 8 3
run s1024_bf16_mx_qb1 BENCH_DTYPE=bf16 TANGO_ATTN_X8_QB=1 -- attention 64 10 1024 8 3
