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
