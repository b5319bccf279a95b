#!/usr/bin/env python
"""Summarise a rocprofv3 `--kernel-trace --stats` run (rocpd sqlite output) into a small text table for
profiles/.  usage: python tools/rocprof_summary.py <results.db> <out.txt> "<command line that was profiled>" """
import sqlite3
import sys


def main():
    db, out, cmd = sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else ""
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                          "group by name order by sum(duration) desc"))
    tot = sum(r[2] for r in rows)
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- %s\n" % cmd)
        f.write("# per-kernel totals from the rocpd `kernels` view (durations in ns there); total kernel time %.3f ms\n" % (tot / 1e6))
        f.write("%-104s %8s %12s %12s %12s %12s %7s\n" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"))
        for n, calls, dur, avg, mn, mx in rows:
            if dur / tot < 0.0003:
                continue
            f.write("%-104s %8d %12.3f %12.1f %12.1f %12.1f %6.2f%%\n" % (n[:104], calls, dur / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * dur / tot))


if __name__ == "__main__":
    main()
