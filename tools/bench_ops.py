#!/usr/bin/env python
"""Micro-benchmark of single engine ops through the C ABI (for rocprofv3 --pmc runs).
usage: python tools/bench_ops.py linear M N K [reps [res|nores [geglu]]]   |   conv B C H W Cout [reps [randn|zeros|ones]]
       python tools/bench_ops.py linear_ln M N K [reps [geglu]]      (LayerNorm folded into the projection: tango_op_linear_ln)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tango_amd import _lib  # noqa: E402

lib = _lib.load()
kind = sys.argv[1]
p = lambda t: C.c_void_p(t.data_ptr())
if kind == "linear":
    M, N, K = [int(v) for v in sys.argv[2:5]]
    reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
    use_res = (sys.argv[6] != "nores") if len(sys.argv) > 6 else True
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    geglu = 1 if len(sys.argv) > 7 and sys.argv[7] == "geglu" else 0
    No = N // 2 if geglu else N
    r = torch.randn(M, No, device="cuda")
    out = torch.empty(M, No, device="cuda")
    for _ in range(reps):
        assert lib.tango_op_linear(1, p(x), p(w), p(b), p(r) if use_res else None, p(out), M, N, K, 0, 0, geglu, None) == 0
elif kind == "linear_ln":
    M, N, K = [int(v) for v in sys.argv[2:5]]
    reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
    geglu = 1 if len(sys.argv) > 6 and sys.argv[6] == "geglu" else 0
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    gam, bet = 1.0 + 0.1 * torch.randn(K, device="cuda"), 0.1 * torch.randn(K, device="cuda")
    out = torch.empty(M, N // 2 if geglu else N, device="cuda")
    for _ in range(reps):
        assert lib.tango_op_linear_ln(1, p(x), p(w), p(b), p(gam), p(bet), None, p(out), M, N, K, geglu, 1e-5, None) == 0
elif kind == "conv":
    B, Cc, H, W, Co = [int(v) for v in sys.argv[2:7]]
    reps = int(sys.argv[7]) if len(sys.argv) > 7 else 5
    fill = sys.argv[8] if len(sys.argv) > 8 else "randn"      # randn | zeros | ones: the DVFS regime probe (MI355X_MICROARCH.md: zero-filled operands clock higher)
    mk = {"randn": torch.randn, "zeros": torch.zeros, "ones": torch.ones}[fill]
    x = mk(B, Cc, H, W, device="cuda")
    w = mk(Co, Cc, 3, 3, device="cuda") / (9 * Cc) ** 0.5
    b = torch.randn(Co, device="cuda")
    out = torch.empty(B, Co, H, W, device="cuda")
    for _ in range(reps):
        assert lib.tango_op_conv2d(1, p(x), p(w), p(b), p(out), B, Cc, H, W, Co, 1, 0, None) == 0
elif kind == "attention":
    # attention B heads S [reps [flags]]   self-attention site of the UNet (d = 64); flags: 0 engine dtype, 1 fp8 P.V, 3 MX fp8 P.V
    B, heads, S = [int(v) for v in sys.argv[2:5]]
    reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
    flags = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    dt = 2 if os.environ.get("BENCH_DTYPE", "fp16") == "bf16" else 1
    q, k, v = (torch.randn(B, S, heads * 64, device="cuda") for _ in range(3))
    out = torch.empty_like(q)
    for _ in range(reps):
        assert lib.tango_op_attention_ex(dt, p(q), p(k), p(v), None, p(out), B, heads, S, S, C.c_float(0.125), flags, None) == 0, lib.tango_last_error()
elif kind == "groupnorm":
    # groupnorm B C rows [reps [act]]   GroupNorm (+ SiLU) over [B, C, rows], 32 groups (norm.hip; TANGO_GN_SLAB=0|1 picks the form for the level 2-3 shapes)
    B, Cc, rows = [int(v) for v in sys.argv[2:5]]
    reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
    act = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    dt = {"fp32": 0, "fp16": 1, "bf16": 2}[os.environ.get("BENCH_DTYPE", "fp16")]
    x = torch.randn(B, Cc, rows, device="cuda")
    ga, be = torch.randn(Cc, device="cuda"), torch.randn(Cc, device="cuda")
    out = torch.empty_like(x)
    for _ in range(reps):
        assert lib.tango_op_groupnorm(dt, p(x), p(ga), p(be), p(out), B, Cc, rows, 32, C.c_float(1e-5), act, None) == 0, lib.tango_last_error()
torch.cuda.synchronize()
print("done")
