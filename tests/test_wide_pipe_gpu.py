"""In-wave software pipeline of the 256 x 320 kernels (round 6: gemm_wide_pipe_kernel, conv3x3_wide_kernel<PIPE>): the fragments of
item i+1 are requested from inline asm under the MFMAs of item i, waited for once per item; DMAs are four items deep.  Same MFMA
order per accumulator and the same epilogues as the ping-pong kernels, so the two must agree to the last bit -- and repeat
bit-identically (the counted waits / barrier placement are hand-written: a race would show as run-to-run differences).

The dispatch switches are launch-time (tuning.h): the tests set the environment, call tango_tuning_reload(), and restore it."""
import ctypes as C
import os

import pytest
import torch
import torch.nn.functional as F

from test_duo_gpu import DT, TOL, p, q, run, tuning

pytestmark = pytest.mark.gpu

REPS = int(os.environ.get("TANGO_PIPE_REPS", "12"))
PIPE = dict(TANGO_WIDE_PIPE=1, TANGO_DUO_MAXK=0, TANGO_FORCE_DMA_GEMM=1)
CONV_PING = dict(TANGO_CONV_PIPE=0, TANGO_CONV_TALL=0, TANGO_FORCE_DMA_GEMM=1)
# 2 = the same pipeline with all eight waves in step and ONE barrier per item (no stagger, no mid-item barrier)
MODES = [1, 2]
PING = dict(TANGO_WIDE_PIPE=0, TANGO_WIDE_PERS=0, TANGO_DUO_MAXK=0, TANGO_FORCE_DMA_GEMM=1)


@pytest.mark.parametrize("pipe", MODES)
@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("M,N,K,res,geglu", [
    (512, 320, 128, 1, 0),          # two tiles, four k-chunks: prologue + peeled last item only
    (768, 640, 160, 0, 0),          # five chunks: one steady-state item with a DMA
    (65536, 640, 640, 1, 0),        # 512 tiles, 20 chunks, residual
    (19200, 2560, 640, 0, 1),       # GEGLU epilogue, uneven rounds
    (16384, 1280, 2560, 0, 0),      # 80 chunks
    (262144, 320, 320, 1, 0),       # the level-0 linears
    (262144, 320, 1280, 1, 0),      # level-0 ff.net.2
])
def test_wide_pipe_linear_bit_equal(lib, pipe, dtype, M, N, K, res, geglu):
    g = torch.Generator().manual_seed(M + N + K + res)
    x = q(torch.randn(M, K, generator=g), dtype).cuda()
    w = q(torch.randn(N, K, generator=g) / K ** 0.5, dtype).cuda()
    b = torch.randn(N, generator=g).cuda()
    No = N // 2 if geglu else N
    r = q(torch.randn(M, No, generator=g), dtype).cuda() if res else None
    call = lambda out: lib.tango_op_linear(DT[dtype], p(x), p(w), p(b), p(r), p(out), M, N, K, 0, 0, geglu, None)
    with tuning(lib, **PING):
        ref = run(lib, call, (M, No))
    with tuning(lib, **dict(PIPE, TANGO_WIDE_PIPE=pipe)):
        out = run(lib, call, (M, No), REPS)
    assert torch.equal(out, ref), "pipelined vs ping-pong 256x320 kernel: %d elements differ" % (out != ref).sum().item()
    h = F.linear(x, w, b)
    if geglu:
        v, gt = h.chunk(2, dim=-1)
        h = v * F.gelu(gt)
    h = h + r if res else h
    err = ((out - h).abs().max() / (h.abs().max() + 1e-9)).item()
    assert err <= TOL[dtype], err


@pytest.mark.parametrize("pipe", MODES)
@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("M,N,K,geglu,res", [(131072, 960, 320, 0, 1), (65536, 1920, 640, 0, 0), (32768, 5120, 640, 1, 0), (16384, 10240, 1280, 1, 0)])
def test_wide_pipe_linear_ln_bit_equal(lib, pipe, dtype, M, N, K, geglu, res):
    """folded LayerNorm: in-loop statistics (plain / residual epilogue) and the external-statistics GEGLU form (levels 1-2)"""
    g = torch.Generator().manual_seed(M + N + K)
    x = q(torch.randn(M, K, generator=g) * 1.3 + 0.7, dtype).cuda()
    w = q(torch.randn(N, K, generator=g) / K ** 0.5, dtype).cuda()
    b = torch.randn(N, generator=g).cuda()
    ga, be = (1 + 0.2 * torch.randn(K, generator=g)).cuda(), (0.3 * torch.randn(K, generator=g)).cuda()
    No = N // 2 if geglu else N
    r = q(torch.randn(M, No, generator=g), dtype).cuda() if res else None
    call = lambda out: lib.tango_op_linear_ln(DT[dtype], p(x), p(w), p(b), p(ga), p(be), p(r), p(out), M, N, K, geglu, C.c_float(1e-5), None)
    with tuning(lib, **PING):
        ref = run(lib, call, (M, No))
    with tuning(lib, **dict(PIPE, TANGO_WIDE_PIPE=pipe)):
        out = run(lib, call, (M, No), REPS)
    assert torch.equal(out, ref), "pipelined vs ping-pong (LN): %d elements differ" % (out != ref).sum().item()


@pytest.mark.parametrize("pipe", MODES)
@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("B,S,Ch,K,ln", [(64, 1024, 640, 640, 1), (64, 1024, 640, 640, 0), (128, 256, 1280, 1280, 1)])
def test_wide_pipe_qkv_vt_bit_equal(lib, pipe, dtype, B, S, Ch, K, ln):
    g = torch.Generator().manual_seed(B + S + Ch + K)
    x = q(torch.randn(B * S, K, generator=g) * 1.2 + 0.4, dtype).cuda()
    w = q(torch.randn(3 * Ch, K, generator=g) / K ** 0.5, dtype).cuda()
    ga, be = (1 + 0.2 * torch.randn(K, generator=g)).cuda(), (0.3 * torch.randn(K, generator=g)).cuda()

    def once():
        oqk = torch.zeros(B * S, 2 * Ch, device="cuda")
        ovt = torch.zeros(B, Ch, S, device="cuda")
        rc = lib.tango_op_linear_qkv(DT[dtype], p(x), p(w), p(ga) if ln else None, p(be) if ln else None, p(oqk), p(ovt), B, S, Ch, K,
                                     C.c_float(1e-5), None)
        assert rc == 0, lib.tango_last_error().decode()
        return oqk, ovt

    with tuning(lib, **PING):
        ref = once()
    with tuning(lib, **dict(PIPE, TANGO_WIDE_PIPE=pipe)):
        for rep in range(REPS):
            o = once()
            assert torch.equal(o[0], ref[0]) and torch.equal(o[1], ref[1]), "repetition %d differs from the ping-pong kernel" % rep


# 3 = the conv's SKEW form: one barrier per item, the halves meet it at different column groups (half an item apart)
@pytest.mark.parametrize("pipe", MODES + [3])
@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("B,Cin,H,W,Cout", [
    (2, 128, 256, 16, 320),         # 32 tiles, four channel chunks (36 items)
    (4, 320, 256, 16, 320),         # level-0 ResBlock conv
    (8, 640, 128, 8, 640),          # level 1: two column tiles
    (16, 1280, 64, 4, 1280),        # level 2
    (64, 1280, 32, 2, 1280),        # level 3: four images per tile
])
def test_wide_pipe_conv_bit_equal(lib, pipe, dtype, B, Cin, H, W, Cout):
    g = torch.Generator().manual_seed(B + Cin + H + Cout)
    x = q(torch.randn(B, Cin, H, W, generator=g), dtype).cuda()
    w = q(torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5, dtype).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    call = lambda out: lib.tango_op_conv2d(DT[dtype], p(x), p(w), p(b), p(out), B, Cin, H, W, Cout, 1, 0, None)
    with tuning(lib, **CONV_PING):
        ref = run(lib, call, (B, Cout, H, W))
    with tuning(lib, **dict(CONV_PING, TANGO_CONV_PIPE=pipe)):
        out = run(lib, call, (B, Cout, H, W), REPS)
    assert torch.equal(out, ref), "pipelined vs ping-pong wide conv: %d elements differ" % (out != ref).sum().item()
    h = F.conv2d(x, w, b, padding=1)
    err = ((out - h).abs().max() / (h.abs().max() + 1e-9)).item()
    assert err <= TOL[dtype], err


@pytest.mark.parametrize("pipe", MODES + [3])
@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("B,Cin,H,W,Cout", [
    (2, 960, 256, 16, 320),         # 32 tiles -> split-K: 30 channel chunks over 7 splits of 5 -- the LAST SPLIT IS EMPTY (Music UNet, B2 = 2)
    (2, 320, 256, 16, 320),         # 10 chunks over 2 splits
    (2, 640, 128, 8, 640),          # 16 tiles, 20 chunks over 5 splits
    (8, 1280, 64, 4, 1280),         # level 2 at B2 = 8: 32 tiles x 8 splits
])
def test_wide_pipe_conv_splitk(lib, pipe, dtype, B, Cin, H, W, Cout):
    """the split-K form of the wide conv as the engine's routing picks it (no TANGO_FORCE_DMA_GEMM: that switch disables split-K), incl. a
    division that leaves the last split without a chunk: the pipelined kernel peels its last item and must not run it on an empty split"""
    g = torch.Generator().manual_seed(B + Cin + H + Cout)
    x = q(torch.randn(B, Cin, H, W, generator=g), dtype).cuda()
    w = q(torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5, dtype).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    call = lambda out: lib.tango_op_conv2d(DT[dtype], p(x), p(w), p(b), p(out), B, Cin, H, W, Cout, 1, 0, None)
    with tuning(lib, TANGO_CONV_PIPE=0, TANGO_CONV_TALL=0):
        ref = run(lib, call, (B, Cout, H, W))
    with tuning(lib, TANGO_CONV_PIPE=pipe, TANGO_CONV_TALL=0):
        out = run(lib, call, (B, Cout, H, W), REPS)
    h = F.conv2d(x, w, b, padding=1)
    err = ((out - h).abs().max() / (h.abs().max() + 1e-9)).item()
    assert err <= TOL[dtype], err
    assert torch.equal(out, ref), "pipelined vs ping-pong wide conv (split-K): %d elements differ" % (out != ref).sum().item()
