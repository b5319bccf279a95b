#!/bin/bash
# gpurun call 3 of round 2: attention / GroupNorm changes, race bisection (DBG variants), post-halo PMC, per-op profile
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "attention or groupnorm or layernorm" > $O/ops_attn.log 2>&1; echo "ops rc=$?"; tail -2 $O/ops_attn.log
TANGO_STRESS_REPS=20 timeout 600 python -m pytest tests/test_determinism_gpu.py -m gpu -q -k "attention or large_mean" > $O/det_attn.log 2>&1; echo "det rc=$?"; tail -3 $O/det_attn.log
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "unet_forward_tiny or denoise_loop_tiny or vae_and_vocoder" > $O/eng.log 2>&1; echo "eng rc=$?"; tail -2 $O/eng.log
for d in 0 1 2 3; do
  TANGO_STREAM_DBG=$d REPS=300 DTYPE=bf16 timeout 200 python tools/diag_stream_race.py > $O/race_dbg$d.txt 2>&1; echo "dbg=$d: $(tail -1 $O/race_dbg$d.txt)"
done
timeout 200 python tools/profile_unet_ops.py --out $O/unet_ops_v22.txt > /dev/null 2>&1; echo "prof rc=$?"; head -1 $O/unet_ops_v22.txt
bash tools/pmc_conv.sh "conv 64 320 256 16 320 4" r2/pmc_halo > $O/pmc_halo.log 2>&1; echo "pmc rc=$?"
python - <<'PY'
import csv,glob,collections,os
out=open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r2/pmc_halo_summary.txt','w')
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r2/pmc_halo/g*/*counter_collection.csv')+glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r2/pmc_halo/*/*/*counter_collection.csv')):
    acc=collections.defaultdict(lambda:[0.0,0])
    for r in csv.DictReader(open(f)):
        if 'conv3x3_halo' in r['Kernel_Name']:
            a=acc[r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1]+=1
    for k,(v,n) in acc.items(): out.write("%s mean_per_launch %.4g launches %d\n"%(k,v/max(n,1),n))
out.close()
print(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r2/pmc_halo_summary.txt').read())
PY
rm -rf $O/pmc_halo/*/*.db $O/pmc_halo/*/*/*.db 2>/dev/null; du -sh $O/pmc_halo
