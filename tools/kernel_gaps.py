#!/usr/bin/env python
"""Kernel-to-kernel gaps from a rocprofv3 `--kernel-trace` run (rocpd sqlite output): how much of a denoise step is spent
BETWEEN kernels.  usage: python tools/kernel_gaps.py <results.db> <out.txt> "<command that was profiled>"

Consecutive dispatches are ordered by start time; gap = start[i+1] - end[i] (negative = overlap, counted as 0).  Gaps above
GAP_CAP_US are host-side pauses (plan building, synchronisation between passes), not dispatch latency, and are left out."""
import sqlite3
import statistics
import sys

GAP_CAP_US = 200.0


def main():
    db, out, cmd = sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else ""
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    start = "start" if "start" in cols else [x for x in cols if "start" in x.lower()][0]
    end = "end" if "end" in cols else [x for x in cols if "end" in x.lower()][0]
    rows = list(c.execute("select name, %s, %s from kernels order by %s" % (start, end, start)))
    gaps, durs = [], []
    by_pred = {}
    for i in range(len(rows) - 1):
        g = (rows[i + 1][1] - rows[i][2]) / 1e3
        durs.append((rows[i][2] - rows[i][1]) / 1e3)
        if g <= GAP_CAP_US:
            g = max(g, 0.0)
            gaps.append(g)
            by_pred.setdefault(rows[i + 1][0], []).append(g)
    with open(out, "w") as f:
        f.write("# kernel-to-kernel gaps, rocprofv3 --kernel-trace -- %s\n" % cmd)
        f.write("# %d dispatches; kernel time %.3f ms; gaps <= %.0f us: %d, total %.3f ms (%.1f %% of kernel time + gaps), "
                "median %.2f us, mean %.2f us, p90 %.2f us\n"
                % (len(rows), sum(durs) / 1e3, GAP_CAP_US, len(gaps), sum(gaps) / 1e3, 100.0 * sum(gaps) / (sum(gaps) + sum(durs) + 1e-9),
                   statistics.median(gaps) if gaps else 0.0, statistics.mean(gaps) if gaps else 0.0,
                   sorted(gaps)[int(0.9 * len(gaps))] if gaps else 0.0))
        f.write("# kernel duration: median %.2f us, mean %.2f us\n" % (statistics.median(durs), statistics.mean(durs)))
        f.write("%-100s %8s %12s %12s\n" % ("gap BEFORE this kernel", "count", "mean_gap_us", "total_ms"))
        for n, g in sorted(by_pred.items(), key=lambda kv: -sum(kv[1]))[:25]:
            f.write("%-100s %8d %12.2f %12.3f\n" % (n[:100], len(g), statistics.mean(g), sum(g) / 1e3))


if __name__ == "__main__":
    main()
