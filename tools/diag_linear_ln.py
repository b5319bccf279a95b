"""Stress / localisation script for the streaming linear with folded LayerNorm (M=5000, N=1920, K=640, fp16).
Runs tango_op_linear_ln REPS times on the same inputs and prints, per repetition, how many outputs deviate from the
torch reference by more than 2 % of max|ref| and where (rows / columns).  Used in round 1 to characterise the
codegen-sensitive race described in DESIGN.md section 5 ("open issue"): a healthy build prints `bad elems 0` every time.
usage (GPU box): REPS=40 python tools/diag_linear_ln.py
"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ctypes as C, torch, torch.nn.functional as F
from tango_amd import _lib
lib = _lib.load()
M, N, K = 5000, 1920, 640
g = torch.Generator().manual_seed(M + N + K)
q = lambda t: t.half().float()
x = q(torch.randn(M, K, generator=g) * 1.3 + 0.7)
w = q(torch.randn(N, K, generator=g) / K ** 0.5)
b = torch.randn(N, generator=g)
ga, be = 1 + 0.2 * torch.randn(K, generator=g), 0.3 * torch.randn(K, generator=g)
h = F.linear(F.layer_norm(x, (K,), ga, be, 1e-5), w, b)
p = lambda t: C.c_void_p(t.data_ptr())
for rep in range(int(os.environ.get("REPS", "3"))):
    out = torch.zeros(M, N, device="cuda")
    xs, ws, bs, gs, es = x.cuda(), w.cuda(), b.cuda(), ga.cuda(), be.cuda()
    rc = lib.tango_op_linear_ln(1, p(xs), p(ws), p(bs), p(gs), p(es), None, p(out), M, N, K, 0, C.c_float(1e-5), None)
    o = out.cpu()
    err = (o - h).abs()
    bad = err > 0.02 * h.abs().max()
    rows = bad.any(1).nonzero().flatten()
    cols = bad.any(0).nonzero().flatten()
    print("rc", rc, "max err", err.max().item(), "bad elems", int(bad.sum()), "rows", rows[:8].tolist(), len(rows), "cols", cols[:8].tolist(), len(cols))
    if len(rows):
        r = rows[0].item()
        print(" row", r, "out", o[r, cols[:4]].tolist(), "ref", h[r, cols[:4]].tolist(), "ratio", (o[r, cols[:4]] / h[r, cols[:4]]).tolist())
