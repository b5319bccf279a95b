#!/bin/bash
# gpurun call 16 of round 2: wide GEMM (own epilogue) -- tests, per-op A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2; mkdir -p $O
timeout 900 python -m pytest tests/test_determinism_gpu.py -m gpu -q -x -k "wide or linear_repeat or geglu_repeat" > $O/det_wide.log 2>&1; echo "det rc=$?"; tail -3 $O/det_wide.log
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "linear" > $O/ops_wide.log 2>&1; echo "ops rc=$?"; tail -1 $O/ops_wide.log
timeout 200 python tools/profile_unet_ops.py --out $O/unet_ops_wide.txt > /dev/null 2>&1; head -1 $O/unet_ops_wide.txt
TANGO_NO_WIDE_GEMM=1 timeout 200 python tools/profile_unet_ops.py --out $O/unet_ops_nowide.txt > /dev/null 2>&1; head -1 $O/unet_ops_nowide.txt
TANGO_WIDE_FIRST=1 timeout 200 python tools/profile_unet_ops.py --out $O/unet_ops_widefirst.txt > /dev/null 2>&1; head -1 $O/unet_ops_widefirst.txt
python - <<'PY'
import re
def load(f):
    d={}
    for l in open(f):
        m=re.match(r"(linear\S* .*?)\s+(\d+)\s+([\d.]+)\s+[\d.]+%",l)
        if m: d[m.group(1).strip()]=(int(m.group(2)),float(m.group(3)))
    return d
a=load("gpurun_out/r2/unet_ops_wide.txt"); b=load("gpurun_out/r2/unet_ops_nowide.txt"); c=load("gpurun_out/r2/unet_ops_widefirst.txt")
print("%-52s %4s %8s %8s %8s"%("op","n","wide","nowide","widefirst"))
for k in sorted(b,key=lambda k:-b[k][1]):
    print("%-52s %4d %8.3f %8.3f %8.3f"%(k,b[k][0],a.get(k,(0,0))[1],b[k][1],c.get(k,(0,0))[1]))
PY
