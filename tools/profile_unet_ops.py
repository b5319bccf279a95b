#!/usr/bin/env python
"""Per-op timing of one eager UNet step (HIP events around every kernel group), aggregated by op label.
usage: python tools/profile_unet_ops.py [--batch 32] [--dtype fp16] [--out file]"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

from tango_amd.engine import UNET_CONFIG_LARGE, Engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--dtype", default="fp16")
ap.add_argument("--out", default=None)
a = ap.parse_args()
e = Engine(unet=UNET_CONFIG_LARGE, dtype=a.dtype)
e.load_synthetic(1234)
rows = e.profile_unet(2 * a.batch, 64)
agg = collections.OrderedDict()
for lab, ms, gf in rows:
    d = agg.setdefault(lab, [0, 0.0, 0.0])
    d[0] += 1
    d[1] += ms
    d[2] += gf
tot = sum(r[1] for r in rows)
lines = ["# one eager UNet step, B=%d (UNet batch %d), L=64, %s: %.2f ms over %d op groups, %.1f GFLOP counted"
         % (a.batch, 2 * a.batch, a.dtype, tot, len(rows), sum(r[2] for r in rows)),
         "%-60s %5s %10s %7s %12s %9s" % ("op", "n", "ms", "pct", "GFLOP", "TFLOP/s")]
for lab, (n, ms, gf) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append("%-60s %5d %10.3f %6.1f%% %12.1f %9.1f" % (lab, n, ms, 100 * ms / tot, gf, gf / ms if ms > 0 else 0))
txt = "\n".join(lines)
print(txt)
if a.out:
    open(a.out, "w").write(txt + "\n")
