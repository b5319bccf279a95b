#!/usr/bin/env python
"""Turn the totals of tools/final_profiles.sh (gpurun_out/final*/pmc_totals.txt: FETCH_SIZE / WRITE_SIZE sums over N = 2 and N = 6 denoise steps) into
the evidence file profiles/<name> and a record of profiles/hbm_traffic.json keyed to the CURRENT kernel sources (bench.kernel_source_sha16).
usage: python tools/update_hbm_traffic.py gpurun_out/final2/pmc_totals.txt profiles/r6_final2_pmc_hbm_traffic_unet_step.txt [src_sha16]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

src, dst = sys.argv[1], sys.argv[2]
tot = {}
lines = [l.strip() for l in open(src) if l.strip()]
for l in lines:
    m = re.match(r"N=(\d+) (\w+) sum ([\d.]+) dispatches (\d+)", l)
    if m:
        tot[(int(m.group(1)), m.group(2))] = float(m.group(3))
fetch = (tot[(6, "FETCH_SIZE")] - tot[(2, "FETCH_SIZE")]) / 4 * 1024
write = (tot[(6, "WRITE_SIZE")] - tot[(2, "WRITE_SIZE")]) / 4 * 1024
sha = sys.argv[3] if len(sys.argv) > 3 else bench.kernel_source_sha16()   # (third argument: the hash the totals were MEASURED on, when it is not the working tree)
with open(os.path.join(ROOT, dst), "w") as o:
    o.write("# HBM-side bytes of one denoise-step launch (B=32, fp16, hipGraph replay) on kernel sources %s\n" % sha)
    o.write("# separate rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE passes of `python bench.py --batch 32 --denoise-steps N --steps 1 --warmup 0 --no-cpu-baseline` (tools/final_profiles.sh),\n")
    o.write("# per step = (sum over N=6 - sum over N=2) / 4; counters in KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 correction)\n")
    for l in lines:
        o.write(l + "\n")
    o.write("fetch raw %.1f GB, write %.1f GB per step -> raw %.1f GB, corrected %.1f GB\n" % (fetch / 1e9, write / 1e9, (fetch + write) / 1e9, (2 * fetch + write) / 1e9))
p = os.path.join(ROOT, "profiles", "hbm_traffic.json")
rec = json.load(open(p))
rec["records"] = [r for r in rec["records"] if r.get("src_sha16") != sha]
rec["records"].append({"batch": 32, "dtype": "fp16", "xl": False, "fp8_attn": False, "src_sha16": sha, "fetch_raw_bytes": fetch, "write_bytes": write,
                       "bytes_per_step": 2 * fetch + write, "bytes_per_step_raw": fetch + write, "source": dst})
json.dump(rec, open(p, "w"), indent=1)
print("record for %s: raw %.1f GB, corrected %.1f GB" % (sha, (fetch + write) / 1e9, (2 * fetch + write) / 1e9))
