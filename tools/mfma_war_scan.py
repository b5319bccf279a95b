#!/usr/bin/env python
"""Static scan of hipcc ISA (-S --cuda-device-only) for writes to the SOURCE registers of an MFMA shortly after it issues.

Round-1 finding ("open issue"): a build of lin_stream_kernel in which hipcc recycled MFMA SrcA
quads for VALU results in the very next instruction produced nondeterministic stale values; pinning the fragments
made it pass 100/100 stress runs.  This tool lists, per kernel, how often a non-MFMA instruction overwrites a register
of an in-flight MFMA's SrcA/SrcB/SrcC within N instructions (MFMA -> MFMA chains are sequenced by the matrix pipe and
are not counted).  Asynchronous writers (ds_read / global_load / scratch_load) are reported separately: their data
lands much later.

usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only file.hip -o file.s
       python tools/mfma_war_scan.py file.s [max_distance=4]
"""
import collections
import re
import sys


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def kernels(path):
    name, body = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name:
            body.append(line.strip())
            if "s_endpgm" in line:
                yield name, body
                name = None


def scan(body, maxd):
    ins = [l for l in body if l and not l.startswith((";", ".")) and not l.endswith(":")]
    sync, asyn = collections.Counter(), collections.Counter()
    for i, l in enumerate(ins):
        if not l.startswith("v_mfma"):
            continue
        ops = [t.strip() for t in l.split(None, 1)[1].split(",")]
        dst = regs(ops[0])
        for label, tok in zip("ABC", ops[1:4]):
            src = regs(tok)
            if not src or src == dst:
                continue
            for j in range(i + 1, min(i + 1 + maxd, len(ins))):
                m = ins[j]
                parts = m.split(None, 1)
                if len(parts) < 2 or m.startswith(("s_", "ds_write", "global_store", "scratch_store", "buffer_store")):
                    continue
                if regs(parts[1].split(",")[0].strip()) & src:
                    if m.startswith("v_mfma"):
                        break
                    op = parts[0]
                    (asyn if op.startswith(("ds_", "global_load", "scratch_load", "buffer_load")) else sync)[(label, j - i)] += 1
                    break
    return sync, asyn, sum(1 for l in ins if l.startswith("v_mfma"))


def main():
    path = sys.argv[1]
    maxd = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    for name, body in kernels(path):
        sync, asyn, nm = scan(body, maxd)
        if not nm:
            continue
        s = ", ".join("Src%s@+%d x%d" % (k[0], k[1], v) for k, v in sorted(sync.items(), key=lambda kv: kv[0][1]))
        a = sum(asyn.values())
        print("%-100s mfma %4d | VALU overwrites within %d: %s | async loads: %d" % (name[:100], nm, maxd, s or "none", a))


if __name__ == "__main__":
    main()
