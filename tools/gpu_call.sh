#!/bin/bash
# gpurun call 21 of round 2: conv on the 256 x 320 tile -- tests, per-op A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2; mkdir -p $O
timeout 900 python -m pytest tests/test_determinism_gpu.py -m gpu -q -x -k "conv3x3" > $O/det_cw.log 2>&1; echo "det rc=$?"; tail -3 $O/det_cw.log; grep -E "^FAILED|rel err|differs" $O/det_cw.log | head
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "conv" > $O/ops_cw.log 2>&1; echo "ops rc=$?"; tail -1 $O/ops_cw.log
timeout 200 python tools/profile_unet_ops.py --out $O/unet_ops_cw.txt > /dev/null 2>&1; head -1 $O/unet_ops_cw.txt
TANGO_NO_WIDE_CONV=1 timeout 200 python tools/profile_unet_ops.py --out $O/unet_ops_nocw.txt > /dev/null 2>&1; head -1 $O/unet_ops_nocw.txt
python - <<'PY'
import re
def load(f):
    d={}
    for l in open(f):
        m=re.match(r"(conv\S* .*?)\s+(\d+)\s+([\d.]+)\s+[\d.]+%\s+[\d.]+\s+([\d.]+)",l)
        if m: d[m.group(1).strip()]=(int(m.group(2)),float(m.group(3)),float(m.group(4)))
    return d
a=load("gpurun_out/r2/unet_ops_cw.txt"); b=load("gpurun_out/r2/unet_ops_nocw.txt")
print("%-52s %4s %8s %8s   TF wide / halo"%("op","n","wide","halo"))
for k in sorted(b,key=lambda k:-b[k][1])[:24]:
    print("%-52s %4d %8.3f %8.3f   %6.0f / %6.0f"%(k,b[k][0],a.get(k,(0,0,0))[1],b[k][1],a.get(k,(0,0,0))[2],b[k][2]))
PY
