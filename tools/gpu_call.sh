#!/bin/bash
# gpurun call 20 of round 2: full GPU suite + smoke + bench on the current tree
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x > $O/full_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/full_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench_v25.json 2> $O/bench_v25.err; cat $O/bench_v25.json
