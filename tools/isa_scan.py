#!/usr/bin/env python
"""Static scan of every gfx950 kernel in libtango_hip.so for the instruction pattern of the round-1/2 streaming-kernel
miscompare.

What failed (profiles/r2_race_hunt.txt): the bf16 K = 640 folded-LayerNorm epilogue, written as plain C++
`rstd * (acc - mean * wsum) + b`, was SLP-vectorised by hipcc 7.2 into

    ds_read_b128 v[210:213], ...                       ; wsum quad from LDS
    v_pk_fma_f32 v[210:211], v[210:211], v[152:153], v[136:137] ... neg_lo:[1,0,0] neg_hi:[1,0,0]   ; elements 0, 1: -wsum*mean + acc
    v_xor_b32    v137, 0x80000000, v213                ; elements 2, 3: sign flip as a separate VALU op ...
    v_xor_b32    v136, 0x80000000, v212                ; ... INTO the registers the packed FMA above still names as a source
    v_pk_fma_f32 v[136:137], v[136:137], v[152:153], v[138:139] op_sel_hi:[1,0,1]

and 39 of 300 bitwise repeats came out with the `mean * wsum` term of elements 2, 3 missing in lanes 48-63.  The hardware
mechanism was never established; the shipped epilogues express the term as ONE `v_fma_f32` in inline asm (0 of 1000).  This
tool is the tripwire for a compiler upgrade or a new epilogue re-creating the form.  It reports, per kernel:

  xor_fed    v_pk_fma_f32 with a source register written, <= 8 instructions earlier, by `v_xor_b32 vX, 0x80000000, vY`
             where vY came out of a ds_read <= 24 instructions earlier                (the failing data flow)
  war_pk     a VALU write, <= 2 instructions after a v_pk_* instruction, to a register that v_pk_* reads as a source
             and does not itself write                                                  (the failing register reuse)

  mfma_raw   (round 3) a non-MFMA instruction that READS the destination of a v_mfma fewer wait states later than the hardware
             needs (s_nop N counts N + 1, every other instruction 1).  hipcc pads these hazards for the instructions it
             models but NOT for inline-asm operands: the fused cross-attention kernel's asm `v_fma_f32` read fresh Q accumulators
             three slots after the last MFMA and its output differed between repetitions until `s_nop`s were added
             (tango_amd/csrc/xattn.hip).  Thresholds are the smallest distances hipcc itself leaves in this library:
             7 for the 4-pass v_mfma_f32_16x16x32_{f16,bf16,fp8}, 10 for the 8-pass v_mfma_f32_16x16x4_f32.

`--check` exits non-zero if any kernel has mfma_raw > 0, or xor_fed > 0 AND war_pk > 0 at the same site (the exact failing shape) or if a
folded-LayerNorm kernel (lin_stream_kernel<.., LN = true, ..>, gemm_wide_kernel<.., LN = true, ..>, xattn / any kernel whose
name is given with --strict) has xor_fed > 0.

usage: python tools/isa_scan.py [--lib tango_amd/lib/libtango_hip.so] [--check] [--out profiles/r3_isa_scan.txt]"""
import argparse
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def regs(tok):
    """set of ('v', n) registers named by one operand token"""
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def parse_inst(line):
    """'\tv_pk_fma_f32 v[1:2], v[3:4], ... mods // addr: enc' -> (mnemonic, [operand tokens], text)"""
    text = line.split("//")[0].strip()
    if not text or text.endswith(":") or text.startswith((".", ";", "<")):
        return None
    parts = text.split(None, 1)
    mn = parts[0]
    ops = []
    if len(parts) > 1:
        # split on commas outside brackets
        depth, cur = 0, ""
        for ch in parts[1]:
            if ch == "[":
                depth += 1
            elif ch == "]":
                depth -= 1
            if ch == "," and depth == 0:
                ops.append(cur.strip())
                cur = ""
            else:
                cur += ch
        if cur.strip():
            ops.append(cur.strip())
    return mn, ops, text


def dst_src(mn, ops):
    """(written registers, read registers) for the instruction classes the scan cares about (VALU / DS reads)"""
    if not ops:
        return set(), set()
    first = ops[0].split()[0] if ops[0] else ""
    if mn.startswith(("ds_read", "ds_load")):
        return regs(first), set()
    if mn.startswith(("v_cmp", "v_cmpx")) or mn.startswith(("s_", "buffer_", "global_", "flat_", "ds_", "scratch_")):
        return set(), set().union(*[regs(o) for o in ops]) if ops else set()
    w = regs(first)
    r = set()
    for o in ops[1:]:
        r |= regs(o.split(" ")[0])
    if mn.endswith(("fmac_f32_e32", "fmac_f32_e64", "fmac_f32")) or mn.startswith("v_mfma") or "mac" in mn:
        r |= w
    return w, r


def scan_kernel(insts):
    xor_fed = war_pk = both = 0
    sites = []
    n = len(insts)
    for i, (mn, ops, text) in enumerate(insts):
        if not mn.startswith("v_pk_"):
            continue
        w, r = dst_src(mn, ops)
        # --- war_pk: a following VALU write into a source of this packed instruction
        war_here = False
        for j in range(i + 1, min(n, i + 3)):
            mn2, ops2, text2 = insts[j]
            if not mn2.startswith("v_") or mn2.startswith(("v_cmp", "v_mfma")):
                continue
            w2, _ = dst_src(mn2, ops2)
            if w2 & (r - w):
                war_here = True
                break
        # --- xor_fed: only for the fused multiply-add form
        xf_here = False
        if mn == "v_pk_fma_f32":
            for j in range(max(0, i - 8), i):
                mn2, ops2, text2 = insts[j]
                if mn2.startswith("v_xor_b32") and "0x80000000" in text2:
                    w2, r2 = dst_src(mn2, ops2)
                    if not (w2 & r):
                        continue
                    for k in range(max(0, j - 24), j):
                        mn3, ops3, _ = insts[k]
                        if mn3.startswith("ds_read") and (dst_src(mn3, ops3)[0] & r2):
                            xf_here = True
                            break
                if xf_here:
                    break
        xor_fed += xf_here
        war_pk += war_here
        if xf_here and war_here:
            both += 1
        if xf_here:
            sites.append(text)
    return xor_fed, war_pk, both, sites


def mfma_need(mn):
    """wait states a non-MFMA reader of this MFMA's destination must be away (empirical floor of compiler-generated code here)"""
    if "_f32_16x16x4_f32" in mn or "32x32" in mn:
        return 10
    return 7


def scan_mfma_raw(insts):
    """(count, example sites) of non-MFMA instructions reading a v_mfma destination too early"""
    n, hits, sites = len(insts), 0, []
    for i, (mn, ops, text) in enumerate(insts):
        if not mn.startswith("v_mfma"):
            continue
        w, _ = dst_src(mn, ops)
        need, ws = mfma_need(mn), 0
        for j in range(i + 1, min(n, i + 24)):
            mn2, ops2, t2 = insts[j]
            if ws >= need or mn2.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc", "s_barrier")):
                break
            w2, r2 = dst_src(mn2, ops2)
            if mn2.startswith("v_mfma"):
                if w2 & w:
                    break                      # accumulated / overwritten in the matrix pipe: a different (interlocked) dependency
                ws += 1
                continue
            named = set().union(*[regs(o) for o in ops2]) if ops2 else set()
            if ((named - w2) | r2) & w:
                hits += 1
                sites.append("%s  ->  %s  (%d wait states, %d needed)" % (text, t2, ws, need))
                break
            if w2 & w:
                break                          # overwritten without being read (WAW is a separate, compiler-handled class)
            ws += (int(ops2[0], 0) + 1) if mn2 == "s_nop" and ops2 else 1
    return hits, sites


def disassemble(lib):
    tmp = tempfile.mkdtemp(prefix="isa_scan_")
    try:
        shutil.copy(lib, os.path.join(tmp, "lib.so"))
        subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        kernels = {}
        for f in sorted(os.listdir(tmp)):
            if "gfx950" not in f:
                continue
            txt = subprocess.run([OBJDUMP, "-d", f], cwd=tmp, check=True, capture_output=True, text=True).stdout
            cur = None
            for line in txt.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    cur = m.group(1)
                    kernels.setdefault(cur, [])
                    continue
                if cur is None:
                    continue
                ins = parse_inst(line)
                if ins:
                    kernels[cur].append(ins)
        return kernels
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return dict(zip(names, out.splitlines()))
    except Exception:
        return {n: n for n in names}


def is_ln_kernel(dm):
    """folded-LayerNorm instantiations: lin_stream_kernel<T, KS, TN, LN = true, ...>, gemm_wide_kernel<T, GEGLU, RES, LN = true, ...>,
    the fused cross-attention block (xattn).  Works on demangled AND mangled names (binutils' c++filt does not know DF16b)."""
    m = re.search(r"lin_stream_kernel<[^,]+, \d+, \d+, (true|false)", dm)
    if m:
        return m.group(1) == "true"
    m = re.search(r"gemm_wide_kernel<[^,]+, (true|false), (true|false), (true|false)", dm)
    if m:
        return m.group(3) == "true"
    m = re.search(r"lin_stream_kernelI(?:f|DF16_|DF16b)Li\d+ELi\d+ELb([01])E", dm)
    if m:
        return m.group(1) == "1"
    m = re.search(r"gemm_wide_kernelI(?:f|DF16_|DF16b)Lb[01]ELb[01]ELb([01])E", dm)
    if m:
        return m.group(1) == "1"
    return "xattn" in dm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "tango_amd", "lib", "libtango_hip.so"))
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    kernels = disassemble(a.lib)
    dm = demangle(list(kernels))
    rows, bad = [], []
    for name, insts in sorted(kernels.items(), key=lambda kv: dm[kv[0]]):
        if not insts:
            continue
        xf, war, both, sites = scan_kernel(insts)
        raw, raw_sites = scan_mfma_raw(insts)
        npk = sum(1 for i in insts if i[0] == "v_pk_fma_f32")
        rows.append((dm[name], len(insts), npk, xf, war, both, raw))
        if both > 0 or (xf > 0 and is_ln_kernel(dm[name])) or raw > 0:
            bad.append((dm[name], xf, war, both, (raw_sites or sites)[:3]))
    lines = ["# tools/isa_scan.py over %s: %d kernels" % (os.path.relpath(a.lib, ROOT), len(rows)),
             "# columns: instructions, v_pk_fma_f32, xor_fed, war_pk, both-at-one-site (the round-2 failing shape), mfma_raw (early read "
             "of an MFMA result: the round-3 inline-asm hazard)   [LN = folded-LayerNorm epilogue]",
             "%8s %8s %8s %8s %6s %8s  %s" % ("insts", "pk_fma", "xor_fed", "war_pk", "both", "mfma_raw", "kernel")]
    for d, n, npk, xf, war, both, raw in rows:
        lines.append("%8d %8d %8d %8d %6d %8d  %s%s" % (n, npk, xf, war, both, raw, "[LN] " if is_ln_kernel(d) else "", d))
    tot = [sum(r[i] for r in rows) for i in (3, 4, 5, 6)]
    lines.append("# totals: xor_fed %d, war_pk %d, both %d, mfma_raw %d; flagged kernels: %d" % (tot[0], tot[1], tot[2], tot[3], len(bad)))
    for d, xf, war, both, sites in bad:
        lines.append("# FLAGGED %s: xor_fed %d war_pk %d both %d e.g. %s" % (d, xf, war, both, sites))
    txt = "\n".join(lines)
    print(txt)
    if a.out:
        open(a.out, "w").write(txt + "\n")
    if a.check and bad:
        sys.exit(1)


if __name__ == "__main__":
    main()
