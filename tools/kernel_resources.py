#!/usr/bin/env python
"""Register / scratch budget of every gfx950 kernel in libtango_hip.so, from the code objects' metadata notes
(`.vgpr_count`, `.vgpr_spill_count`, `.private_segment_fixed_size`).

Why (round 5): the level-0 GEGLU projection's streaming kernel carried 21 spilled VGPRs -- scratch reloads retire through the same
in-order vmcnt queue as its operand prefetch -- and a generic epilogue; specialising it (SPEC = 1) removed both and made the op 15 %
faster.  Spills in a hot kernel are invisible in the source and cheap to detect here; tests/test_host_logic.py pins the hot kernels at zero.

usage: python tools/kernel_resources.py [--lib tango_amd/lib/libtango_hip.so] [--out file]"""
import argparse
import os
import re
import shutil
import subprocess
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_resources(lib):
    """{mangled kernel name: (vgprs, spilled vgprs, scratch bytes per lane)}"""
    tmp = tempfile.mkdtemp()
    try:
        shutil.copy(lib, os.path.join(tmp, "lib.so"))
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=tmp, check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        res = {}
        for f in sorted(os.listdir(tmp)):
            if "gfx950" not in f:
                continue
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f], cwd=tmp, check=True, capture_output=True, text=True).stdout
            for blk in re.split(r"- \.agpr_count", txt)[1:]:
                n = re.search(r"\.name:\s+(\S+)", blk)
                sc = re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk)
                vg = re.search(r"\.vgpr_count:\s+(\d+)", blk)
                sp = re.search(r"\.vgpr_spill_count:\s+(\d+)", blk)
                if n and sc and vg:
                    res[n.group(1)] = (int(vg.group(1)), int(sp.group(1)) if sp else 0, int(sc.group(1)))
        return res
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def template_args(mangled):
    """('kernel_name', [template arguments]) of an Itanium-mangled `tango::name<...>(...)` kernel symbol -- only the argument kinds
    this library's kernels use: types f / DF16_ / DF16b -> 'f32' / 'f16' / 'bf16', Lb0E / Lb1E -> bool, Li<n>E / Lin<n>E -> int.
    (binutils' c++filt does not know DF16_; llvm-cxxfilt is not in this image.)  Returns (name, None) for anything else."""
    m = re.match(r"_ZN5tango(\d+)", mangled)
    if not m:
        return mangled, None
    n = int(m.group(1))
    start = m.end()
    name, rest = mangled[start:start + n], mangled[start + n:]
    if not rest.startswith("I"):
        return name, []
    rest, args = rest[1:], []
    while rest and not rest.startswith("E"):
        mm = re.match(r"DF16_|DF16b|f|Lb([01])E|Li(n?)(\d+)E", rest)
        if not mm:
            return name, None
        tok = mm.group(0)
        if tok == "DF16_": args.append("f16")
        elif tok == "DF16b": args.append("bf16")
        elif tok == "f": args.append("f32")
        elif tok.startswith("Lb"): args.append(mm.group(1) == "1")
        else: args.append(-int(mm.group(3)) if mm.group(2) else int(mm.group(3)))
        rest = rest[len(tok):]
    return name, args


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "tango_amd", "lib", "libtango_hip.so"))
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    res = kernel_resources(a.lib)
    lines = ["# %d kernels in %s; kernels with scratch:" % (len(res), os.path.relpath(a.lib, ROOT)),
             "%-110s %6s %8s %8s" % ("kernel", "VGPRs", "spilled", "scratch")]
    for k, (vg, sp, sc) in sorted(res.items(), key=lambda kv: -kv[1][2]):
        if sc > 0:
            lines.append("%-110s %6d %8d %8d" % (k[:110], vg, sp, sc))
    txt = "\n".join(lines) + "\n"
    if a.out:
        open(a.out, "w").write(txt)
    print(txt, end="")


if __name__ == "__main__":
    main()
