#!/usr/bin/env python
"""Per-op timing of one eager UNet step (HIP events around every kernel group), aggregated by op label.
usage: python tools/profile_unet_ops.py [--batch 32] [--dtype fp16] [--out file]
       python tools/profile_unet_ops.py --ab "TANGO_WIDE_SCHED=0;TANGO_WIDE_SCHED=2" [--rounds 3] [--grep conv3x3]
         A/B inside ONE process: the arms (';'-separated, each a ','-separated list of ENV=VALUE launch-time switches) are
         timed round-robin, `tango_tuning_reload()` between them; per op label the MEDIAN over the rounds is reported per arm."""
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
ap.add_argument("--ab", default=None)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--grep", default=None, help="only print op labels containing one of these '|'-separated substrings (totals are always printed)")
a = ap.parse_args()
e = Engine(unet=UNET_CONFIG_LARGE, dtype=a.dtype)
e.load_synthetic(1234)
if a.ab:
    import statistics
    arms = [x.strip() for x in a.ab.split(";") if x.strip()]
    data = {arm: collections.OrderedDict() for arm in arms}      # arm -> label -> [per-round ms]
    totals = {arm: [] for arm in arms}
    counts = {}
    for r in range(a.rounds):
        for arm in arms:
            saved = {}
            for kv in arm.split(","):
                k, v = kv.split("=")
                saved[k] = os.environ.get(k)
                os.environ[k] = v
            e.lib.tango_tuning_reload()
            e.drop_plans()          # routing decisions taken when a plan is built (split-K, folded LayerNorm or not, labels) follow the arm
            rows = e.profile_unet(2 * a.batch, 64)
            for k, v in saved.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v
            per = collections.OrderedDict()
            for lab, ms, gf in rows:
                d = per.setdefault(lab, [0, 0.0, 0.0])
                d[0] += 1; d[1] += ms; d[2] += gf
            for lab, (n, ms, gf) in per.items():
                data[arm].setdefault(lab, []).append(ms)
                counts[lab] = (n, gf)
            totals[arm].append(sum(x[1] for x in rows))
    e.lib.tango_tuning_reload()
    lines = ["# A/B of one eager UNet step, B=%d, %s, %d interleaved rounds, median ms per op label (TFLOP/s)" % (a.batch, a.dtype, a.rounds),
             "%-52s %4s " % ("op", "n") + " ".join("%24s" % arm[-24:] for arm in arms)]
    lines.append("%-52s %4s " % ("TOTAL step", "") + " ".join("%24s" % ("%.2f" % statistics.median(totals[arm])) for arm in arms))
    labs = sorted(counts, key=lambda l: -statistics.median(data[arms[0]].get(l, [0.0])))
    fams = collections.OrderedDict()
    for lab in labs:
        fam = lab.split(" ")[0]
        for arm in arms:
            fams.setdefault(fam, {}).setdefault(arm, 0.0)
            fams[fam][arm] += statistics.median(data[arm].get(lab, [0.0]))
    for fam, d in fams.items():
        lines.append("%-52s %4s " % ("family " + fam, "") + " ".join("%24s" % ("%.3f" % d[arm]) for arm in arms))
    pats = a.grep.split("|") if a.grep else None
    for lab in labs:
        if pats and not any(p in lab for p in pats):
            continue
        n, gf = counts[lab]
        cells = []
        for arm in arms:
            ms = statistics.median(data[arm].get(lab, [0.0]))
            cells.append("%24s" % ("%.3f (%4.0f)" % (ms, gf / ms if ms > 0 else 0)))
        lines.append("%-52s %4d " % (lab[:52], n) + " ".join(cells))
    txt = "\n".join(lines)
    print(txt)
    if a.out:
        open(a.out, "w").write(txt + "\n")
    sys.exit(0)
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
