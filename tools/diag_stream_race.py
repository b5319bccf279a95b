"""Characterise the nondeterministic miscompare of the streaming linear + folded LayerNorm.

Runs tango_op_linear_ln REPS times on identical inputs (default: the shape that fails, M=5000 N=1920 K=640) and, for every
repetition whose output differs from the majority result, prints WHERE (rows / columns, position inside the 32-row group and
the 80-column panel) and WHAT the wrong values look like next to candidate explanations:
  * `nolast`  : the same output with the last 64-byte k-step of x dropped (a missed final MFMA)
  * `prevgrp` : the value 256 rows earlier in the same column (the wave's previous 32-row group: a stale accumulator)
  * `bias`    : the epilogue constants only (accumulator read as zero)
usage (GPU box): REPS=300 DTYPE=bf16 python tools/diag_stream_race.py   # TANGO_STREAM_NOFIX=1: the build without the vmcnt(0) fix
                 # (reproduces 35 / 300); TANGO_NO_STAGED_EPILOGUE=1: direct stores
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from tango_amd import _lib  # noqa: E402

lib = _lib.load()
M, N, K = [int(v) for v in os.environ.get("SHAPE", "5000,1920,640").split(",")]
dtype = os.environ.get("DTYPE", "bf16")
DT = {"fp32": 0, "fp16": 1, "bf16": 2}[dtype]
reps = int(os.environ.get("REPS", "200"))
g = torch.Generator().manual_seed(M + N + K)
q = (lambda t: t.half().float()) if dtype == "fp16" else (lambda t: t.bfloat16().float()) if dtype == "bf16" else (lambda t: t)
x = q(torch.randn(M, K, generator=g) * 1.3 + 0.7).cuda()
w = q(torch.randn(N, K, generator=g) / K ** 0.5).cuda()
b = torch.randn(N, generator=g).cuda()
ga, be = (1 + 0.2 * torch.randn(K, generator=g)).cuda(), (0.3 * torch.randn(K, generator=g)).cuda()
p = lambda t: C.c_void_p(t.data_ptr())


PLAIN = os.environ.get("PLAIN") == "1"        # PLAIN=1: the same kernel family without the folded LayerNorm (bias epilogue only)


def run():
    out = torch.zeros(M, N, device="cuda")
    if PLAIN:
        rc = lib.tango_op_linear(DT, p(x), p(w), p(b), None, p(out), M, N, K, 0, 0, 0, None)
    else:
        rc = lib.tango_op_linear_ln(DT, p(x), p(w), p(b), p(ga), p(be), None, p(out), M, N, K, 0, C.c_float(1e-5), None)
    assert rc == 0, lib.tango_last_error().decode()
    return out


ref = F.linear(x, w, b) if PLAIN else F.linear(F.layer_norm(x, (K,), ga, be, 1e-5), w, b)
outs = [run() for _ in range(3)]
good = outs[0] if torch.equal(outs[0], outs[1]) or torch.equal(outs[0], outs[2]) else outs[1]
if PLAIN:
    ga, be = torch.ones_like(ga), torch.zeros_like(be)
print("shape M=%d N=%d K=%d %s%s; majority result rel err vs reference %.3e" % (M, N, K, dtype, " PLAIN" if PLAIN else "", ((good - ref).abs().max() / ref.abs().max()).item()))
# candidate explanations, computed with torch from the same folded quantities the kernel uses
mu = x.mean(1, keepdim=True)
rstd = (x.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
wf = q(w * ga)                                     # W' as stored
bfold = b + w @ be
wsum = wf.sum(1)
kel = 64 // (2 if dtype != "fp32" else 4)           # elements in the last 64-byte k-step
acc_nolast = x[:, :K - kel] @ wf[:, :K - kel].t()
nolast = rstd * (acc_nolast - mu * wsum) + bfold
biasonly = rstd * (0 - mu * wsum) + bfold
nowsum = rstd * (x @ wf.t()) + bfold                 # the wsum constant read as 0 (what round 2 found)
nbad = 0
for rep in range(reps):
    out = run()
    if torch.equal(out, good):
        continue
    nbad += 1
    bad = (out != good).nonzero()
    rows, cols = bad[:, 0], bad[:, 1]
    r0, c0 = int(rows[0]), int(cols[0])
    print("rep %d: %d differing elements; rows %d..%d (%d distinct), cols %s; row %% 32 = %d, group %d, col %% 80 = %d (panel %d)"
          % (rep, bad.shape[0], int(rows.min()), int(rows.max()), rows.unique().numel(), cols.unique().tolist()[:6], r0 % 32, r0 // 32,
             c0 % 80, c0 // 80))
    for (r, c) in bad[:4].tolist():
        prev = good[r - 256, c].item() if r >= 256 else float("nan")
        print("   [%d,%d] bad % .5f good % .5f ref % .5f | nolast % .5f prevgrp % .5f bias-only % .5f wsum-read-as-0 % .5f | bad-good % .5f"
              % (r, c, out[r, c].item(), good[r, c].item(), ref[r, c].item(), nolast[r, c].item(), prev, biasonly[r, c].item(),
                 nowsum[r, c].item(), (out[r, c] - good[r, c]).item()))
    # does the deficit equal ONE k-step's contribution (a single MFMA of the chain lost / fed a stale operand)?
    (r, c) = bad[0].tolist()
    d = (out[r, c] - good[r, c]).item()
    contrib = torch.stack([rstd[r, 0] * (x[r, j * kel:(j + 1) * kel] @ wf[c, j * kel:(j + 1) * kel]) for j in range(K // kel)])
    j = int((contrib + d).abs().argmin())
    print("   deficit % .5f; closest single k-step contribution: step %d of %d: % .5f (all steps: %s)"
          % (d, j, K // kel, contrib[j].item(), ["%.3f" % v for v in contrib.tolist()]))
print("%d of %d repetitions differed" % (nbad, reps))
