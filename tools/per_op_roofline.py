#!/usr/bin/env python
"""Per-op roofline of one UNet step: for every row of a tools/profile_unet_ops.py table, the algorithmic FLOP (from the table) and
the minimum HBM bytes of that op (inputs once + weights once + outputs once, 16-bit activations), the floor time
max(FLOP / 2.5 PFLOP/s, bytes / 8 TB/s) (MI355X_MICROARCH.md peaks: dense 16-bit MFMA, HBM3E), the measured time and their ratio.

usage: python tools/per_op_roofline.py profiles/r3_final_unet_step_per_op_fp16_b32.txt [--batch2 64] [--out file]

Byte model (a LOWER bound: residual reads, bias / scale vectors and halo re-reads are not counted):
  linear M N K          A = M K, W = N K, out = M N  (GEGLU projections -- N = 8 C -- write M N / 2)
  conv3x3 M N K         Cin = K / 9: A = M Cin (each input pixel once), W = N K, out = M N;  up: A = M Cin / 4;  s2: A = 4 M Cin
  attention Sq Skv h    per sample: q, out = Sq h 64, k, v = Skv h 64
  xattn_block M C       x in, y out (K / V / weights are L2-resident)
  groupnorm C rows      read + write of batch2 x rows x C (the statistics pass is NOT in the floor: it is fusable in principle)
  layernorm C           read + write; rows = batch2 x (1024 for C = 640, 256 for C = 1280)
  ff_fused M C          round 6: x in, y out, weights 12 C^2 once (the [M, 4C] GEGLU tensor does not exist)
  qkv_stat M N K        round 6: as linear
  groupnorm(stats)      round 6: one read of batch2 x rows x C
"""
import argparse
import re

PEAK_FLOPS = 2.5e15
PEAK_HBM = 8.0e12


def op_bytes(name, b2):
    g = lambda k: int(re.search(r"\b%s=(\d+)" % k, name).group(1))  # noqa: E731
    if name.startswith("ff_fused"):
        return 2 * (2 * g("M") * g("C") + 12 * g("C") * g("C"))
    if name.startswith("groupnorm(stats)"):
        return 2 * b2 * g("rows") * g("C")
    if name.startswith(("linear", "conv3x3", "qkv_stat")):
        M, N, K = g("M"), g("N"), g("K")
        out_n = N // 2 if (name.startswith("linear") and N in (2560, 5120, 10240)) else N
        if name.startswith("conv3x3up"):
            a = M * (K // 9) // 4
        elif name.startswith("conv3x3s2"):
            a = 4 * M * (K // 9)
        elif name.startswith("conv3x3"):
            a = M * (K // 9)
        else:
            a = M * K
        return 2 * (a + N * K + M * out_n)
    if name.startswith("attention"):
        sq, skv, h = g("Sq"), g("Skv"), g("heads")
        return 2 * b2 * h * 64 * (2 * sq + 2 * skv)
    if name.startswith("xattn_block"):
        return 2 * 2 * g("M") * g("C")
    if name.startswith("groupnorm"):
        return 2 * 2 * b2 * g("rows") * g("C")
    if name.startswith("layernorm"):
        c = g("C")
        return 2 * 2 * b2 * (1024 if c == 640 else 256 if c == 1280 else 4096) * c
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("table")
    ap.add_argument("--batch2", type=int, default=64, help="UNet batch (2 x prompts)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    rows = []
    head = ""
    for line in open(a.table):
        if line.startswith("#"):
            head = line.strip()
            continue
        m = re.match(r"(.+?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)%\s+([\d.]+)\s+([\d.]+)\s*$", line)
        if not m:
            continue
        name, n, ms, gf = m.group(1).strip(), int(m.group(2)), float(m.group(3)), float(m.group(5))
        by = op_bytes(name, a.batch2) * n
        t_f, t_b = gf * 1e9 / PEAK_FLOPS * 1e3, by / PEAK_HBM * 1e3
        rows.append((name, n, ms, gf, by, t_f, t_b, max(t_f, t_b)))
    tot_ms, tot_floor = sum(r[2] for r in rows), sum(r[7] for r in rows)
    out = ["# per-op roofline of: %s" % head.lstrip("# "),
           "# floor = max(GFLOP / 2500 TFLOP/s, min bytes / 8 TB/s) per op; 'x floor' = measured / floor; 'headroom' = measured - floor (ms per step)",
           "%-58s %3s %8s %8s %8s %8s %6s %7s %8s" % ("op", "n", "ms", "GB min", "t_mfma", "t_hbm", "bound", "x floor", "headroom")]
    for name, n, ms, gf, by, t_f, t_b, fl in sorted(rows, key=lambda r: -(r[2] - r[7])):
        out.append("%-58s %3d %8.3f %8.2f %8.3f %8.3f %6s %7.2f %8.3f" % (name[:58], n, ms, by / 1e9, t_f, t_b, "mfma" if t_f >= t_b else "hbm",
                                                                             ms / fl if fl > 0 else float("nan"), ms - fl))
    out.append("# step: measured %.2f ms, sum of per-op floors %.2f ms (%.1f %% of measured); min bytes %.1f GB, %.1f TFLOP"
               % (tot_ms, tot_floor, 100 * tot_floor / tot_ms, sum(r[4] for r in rows) / 1e9, sum(r[3] for r in rows) / 1e3))
    for fam in ("conv3x3", "linear", "ff_fused", "qkv_stat", "attention", "xattn", "groupnorm", "layernorm"):
        fr = [r for r in rows if r[0].startswith(fam)]
        if fr:
            m_, f_ = sum(r[2] for r in fr), sum(r[7] for r in fr)
            out.append("#   %-10s measured %6.2f ms, floor %6.2f ms (x %.2f)" % (fam, m_, f_, m_ / f_))
    txt = "\n".join(out)
    print(txt)
    if a.out:
        open(a.out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
