"""Deferred rescale of the online softmax (round 6: attn_kernel<..., DEFER>, unmasked 16-bit self-attention sites): the running maximum
moves only when a tile exceeds it by more than 2^8, in between the tile's weights are taken against the OLD maximum (<= 2^8 instead of
<= 1).  The branch that does move it is rare on bounded random data, so (cdna_hip_programming.md rule 26):
  * a FULL-tensor fp64 reference, not bitwise-vs-self;
  * inputs that FORCE the branch -- one key spiked against one query so that the row's maximum jumps by far more than 2^8 at a chosen
    LATE tile, and a staircase of keys that raises the maximum by ~2^3 per tile (the deferred offset accumulates up to the threshold,
    then moves);
  * the exact form (TANGO_ATTN_DEFER=0) and the deferred form must agree to rounding on all of them."""
import ctypes as C

import pytest
import torch

from test_duo_gpu import tuning

pytestmark = pytest.mark.gpu

DT = {"fp16": 1, "bf16": 2}
TOL = {"fp16": 2.5e-3, "bf16": 2e-2}


def q16(t, dtype):
    return t.half().float() if dtype == "fp16" else t.bfloat16().float()


def _run(lib, dtype, q, k, v, B, heads, S, defer):
    out = torch.empty(B, S, heads * 64, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
    with tuning(lib, TANGO_ATTN_DEFER=defer):
        rc = lib.tango_op_attention(DT[dtype], p(q), p(k), p(v), None, p(out), B, heads, S, S, C.c_float(0.125), None)
    assert rc == 0, lib.tango_last_error().decode()
    return out.cpu()


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("case", ["random", "spike_late", "staircase", "spike_first_tile"])
@pytest.mark.parametrize("B,heads,S", [(2, 5, 4096), (3, 10, 1024), (2, 3, 256)])
def test_attention_deferred_rescale(lib, dtype, case, B, heads, S):
    g = torch.Generator().manual_seed(S + heads + len(case))
    C_ = heads * 64
    q = q16(torch.randn(B, S, C_, generator=g), dtype)
    k = q16(torch.randn(B, S, C_, generator=g), dtype)
    v = q16(torch.randn(B, S, C_, generator=g) * 1.5 + 0.3, dtype)
    qh = q.view(B, S, heads, 64)
    kh = k.view(B, S, heads, 64)
    if case in ("spike_late", "spike_first_tile"):
        # key `kk` = 6 x query `qq` of head 0: its logit is 6 |q|^2 / 8 ~ 48 nats = 69 in the exp2 domain, far beyond 2^8 above the
        # row's other scores (max ~ 4); every other row sees an ordinary key
        kk = (S - 70) if case == "spike_late" else 5
        for b in range(B):
            qq = 17 + b
            kh[b, kk, 0] = q16(6.0 * qh[b, qq, 0], dtype)
    elif case == "staircase":
        # keys 64 t + 3 (one per 64-key tile) aligned with query 9 of head 1 at growing gain: the row maximum rises ~2.5 (exp2 domain) per tile
        for t in range(min(S // 64, 24)):
            kh[0, 64 * t + 3, 1] = q16((0.15 + 0.22 * t) * qh[0, 9, 1], dtype)
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    ref = torch.softmax((qh.transpose(1, 2).double() @ kh.transpose(1, 2).double().transpose(-1, -2)) * 0.125, -1) @ v.view(B, S, heads, 64).transpose(1, 2).double()
    ref = ref.transpose(1, 2).reshape(B, S, C_).float()
    exact = _run(lib, dtype, qd, kd, vd, B, heads, S, 0)
    defer = _run(lib, dtype, qd, kd, vd, B, heads, S, 1)
    assert torch.isfinite(defer).all()
    scale = ref.abs().max().item()
    e_exact = (exact - ref).abs().max().item() / scale
    e_defer = (defer - ref).abs().max().item() / scale
    print("attention %s %s B=%d h=%d S=%d: exact-lazy vs fp64 %.3e, deferred vs fp64 %.3e, deferred vs exact-lazy %.3e"
          % (dtype, case, B, heads, S, e_exact, e_defer, (defer - exact).abs().max().item() / scale))
    assert e_exact <= TOL[dtype] and e_defer <= TOL[dtype], (e_exact, e_defer)
    if case == "random" and S >= 1024:
        assert not torch.equal(defer, exact)         # the switch is really on (later tiles use the old offset)
    # repeat: bit-identical
    again = _run(lib, dtype, qd, kd, vd, B, heads, S, 1)
    assert torch.equal(again, defer)


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("B,heads,S", [(2, 5, 4096), (3, 10, 1024), (1, 5, 576)])
def test_attention_kv_tiles_by_lds_dma(lib, dtype, B, heads, S):
    """attn_kernel<..., KDMA, VDMA> at the op level (reference op: attention_processor.py:495-540): a V^T whose keys sit in the kernel's fragment
    order inside every block of 32 (tango_op_attention_ex flags bit 2, AttnParams::vt_perm) is fetched by LDS-DMA like K; same arithmetic in the same
    order as the register-staged kernel -- bitwise equal to it, and against the full fp64 reference"""
    g = torch.Generator().manual_seed(S + heads)
    C_ = heads * 64
    q = q16(torch.randn(B, S, C_, generator=g), dtype)
    k = q16(torch.randn(B, S, C_, generator=g), dtype)
    v = q16(torch.randn(B, S, C_, generator=g) * 1.5 + 0.3, dtype)
    s = torch.arange(S)
    pos = (s & ~31) | ((s & 12) << 1) | ((s & 16) >> 2) | (s & 3)
    vp = torch.empty_like(v)
    vp[:, pos] = v                      # row pos(s) holds key s
    qh, kh, vh = (t.view(B, S, heads, 64).transpose(1, 2).double() for t in (q, k, v))
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * 0.125, -1) @ vh).transpose(1, 2).reshape(B, S, C_).float()
    p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
    outs = {}
    for name, kd, flags, vv in (("staged", 0, 0, v), ("kdma", 1, 0, v), ("kvdma", 1, 4, vp)):
        out = torch.empty(B, S, C_, device="cuda")
        qd, kdv, vd = q.cuda(), k.cuda(), vv.cuda()
        with tuning(lib, TANGO_ATTN_KDMA=kd, TANGO_ATTN_DEFER=0):
            rc = lib.tango_op_attention_ex(DT[dtype], p(qd), p(kdv), p(vd), None, p(out), B, heads, S, S, C.c_float(0.125), flags, None)
        assert rc == 0, lib.tango_last_error().decode()
        outs[name] = out.cpu()
    err = (outs["kvdma"] - ref).abs().max().item() / ref.abs().max().item()
    print("attention %s B=%d heads=%d S=%d: K / V^T tiles by LDS-DMA vs fp64 %.3e" % (dtype, B, heads, S, err))
    assert err <= TOL[dtype]
    assert torch.equal(outs["staged"], outs["kdma"])
    assert torch.equal(outs["staged"], outs["kvdma"])


def test_attention_vt_perm_argument_errors(lib):
    p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
    q = torch.zeros(1, 256, 320, device="cuda")
    out = torch.empty_like(q)
    # a permuted V^T is only understood by the LDS-DMA form (Sq > 512): refused, never read with another kernel
    assert lib.tango_op_attention_ex(1, p(q), p(q), p(q), None, p(out), 1, 5, 256, 256, C.c_float(0.125), 4, None) != 0
    assert b"vt_perm" in lib.tango_last_error()
