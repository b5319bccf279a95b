#!/usr/bin/env python
"""Per-op timing of the two stages that follow the denoise loop in every pass: the mel-VAE decoder and HiFi-GAN (HIP events around
every kernel group of the eager plans: tango_engine_profile_vae / _vocoder), aggregated by op label, with GFLOP and TFLOP/s.
usage: python tools/profile_vae_vocoder_ops.py [--batch 32] [--dtype fp16] [--out file] [--rounds 3]"""
import argparse
import collections
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

from tango_amd.engine import HIFIGAN_CONFIG, VAE_CONFIG, Engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--dtype", default="fp16")
ap.add_argument("--out", default=None)
ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()
e = Engine(vae=VAE_CONFIG, hifigan=HIFIGAN_CONFIG, dtype=a.dtype)
e.load_synthetic(1234)
lines = []
for name, fn in (("mel-VAE decoder", lambda: e.profile_vae(a.batch)), ("HiFi-GAN", lambda: e.profile_vocoder(a.batch, 1024))):
    per = collections.OrderedDict()     # label -> [count, [ms per round], gflop]
    totals = []
    for r in range(a.rounds):
        rows = fn()
        totals.append(sum(x[1] for x in rows))
        acc = collections.OrderedDict()
        for lab, ms, gf in rows:
            d = acc.setdefault(lab, [0, 0.0, 0.0])
            d[0] += 1; d[1] += ms; d[2] += gf
        for lab, (n, ms, gf) in acc.items():
            d = per.setdefault(lab, [n, [], gf])
            d[1].append(ms)
    tot = statistics.median(totals)
    gtot = sum(d[2] for d in per.values())
    lines.append("# %s, B=%d, %s: one eager pass, median of %d rounds: %.3f ms, %.1f GFLOP executed = %.0f TFLOP/s (%.1f %% of the 2.5 PF dense 16-bit MFMA peak)"
                 % (name, a.batch, a.dtype, a.rounds, tot, gtot, gtot / tot if tot else 0.0, 100.0 * gtot / tot / 2500.0 if tot else 0.0))
    lines.append("%-64s %4s %10s %7s %10s %8s" % ("op", "n", "ms", "%", "GFLOP", "TFLOP/s"))
    for lab, (n, mss, gf) in sorted(per.items(), key=lambda kv: -statistics.median(kv[1][1])):
        ms = statistics.median(mss)
        lines.append("%-64s %4d %10.3f %6.1f%% %10.1f %8.0f" % (lab[:64], n, ms, 100.0 * ms / tot, gf, gf / ms if ms > 0 else 0.0))
    lines.append("")
txt = "\n".join(lines)
if a.out:
    os.makedirs(os.path.dirname(os.path.abspath(a.out)) or ".", exist_ok=True)
    open(a.out, "w").write(txt)
print(txt)
