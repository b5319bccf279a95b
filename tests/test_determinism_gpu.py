"""Repeat-run determinism of every MFMA kernel family (VERDICT r1 item 2 / ADVICE r1): the same launch on the same
inputs REPS times, every output bit-identical to the first and equal to the torch reference within the op tolerance.
A single passing run says nothing about a race; the round-1 templated streaming kernel failed 2 of 3 such repetitions
at M=5000 N=1920 K=640 before the hazard work recorded in DESIGN.md section 5.

Also here: folded-LayerNorm numerics on rows whose mean dwarfs their spread (cancellation in one-pass
statistics), and the wide-tile LDS-DMA GEMM at shapes that reach it without any env switch.
"""
import ctypes as C
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = {"fp32": 0, "fp16": 1, "bf16": 2}
TOL = {"fp32": 2e-5, "fp16": 4e-3, "bf16": 3e-2}
REPS = int(os.environ.get("TANGO_STRESS_REPS", "50"))


def q(t, dtype):
    return t.half().float() if dtype == "fp16" else t.bfloat16().float() if dtype == "bf16" else t


def p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


#: the 16-bit K = 640 folded-LayerNorm build of the streaming kernel is the family that miscompared in rounds 1-2 (39 of 300
#: with the un-patched epilogue, profiles/r2_race_hunt.txt): it always runs at >= 300 repetitions
REPS_LN640 = max(REPS, 300)


def repeat(lib, call, out_shape, ref, tol, what, reps=None):
    first = None
    for rep in range(reps or REPS):
        out = torch.zeros(out_shape, device="cuda")
        rc = call(out)
        assert rc == 0, lib.tango_last_error().decode()
        if first is None:
            first = out
            err = ((out.cpu() - ref).abs().max() / (ref.abs().max() + 1e-9)).item()
            assert err <= tol, "%s: rel err %.3e" % (what, err)
        else:
            same = torch.equal(out, first)
            if not same:
                bad = (out != first).nonzero()
                raise AssertionError("%s: repetition %d differs from repetition 0 at %d elements, first %s" %
                                     (what, rep, bad.shape[0], bad[0].tolist()))


@pytest.mark.parametrize("dtype", ["fp16", "bf16", "fp32"])
@pytest.mark.parametrize("M,N,K,geglu,res", [(5000, 1920, 640, 0, 0), (4128, 320, 320, 0, 1), (4100, 2560, 320, 1, 0), (8192, 640, 640, 0, 1)])
def test_stream_linear_ln_repeat(lib, dtype, M, N, K, geglu, res):
    if dtype == "fp32" and K == 640:
        pytest.skip("fp32 rows of 2560 bytes take the LN-kernel + GEMM fallback (covered by test_ops_gpu)")
    g = torch.Generator().manual_seed(M + N + K)
    x = q(torch.randn(M, K, generator=g) * 1.3 + 0.7, dtype).cuda()
    w = q(torch.randn(N, K, generator=g) / K ** 0.5, dtype).cuda()
    b = torch.randn(N, generator=g).cuda()
    ga, be = (1 + 0.2 * torch.randn(K, generator=g)).cuda(), (0.3 * torch.randn(K, generator=g)).cuda()
    No = N // 2 if geglu else N
    r = q(torch.randn(M, No, generator=g), dtype).cuda() if res else None
    h = F.linear(F.layer_norm(x, (K,), ga, be, 1e-5), w, b)
    if geglu:
        v, gt = h.chunk(2, dim=-1)
        h = v * F.gelu(gt)
    ref = (h + r if res else h).cpu()
    repeat(lib, lambda out: lib.tango_op_linear_ln(DT[dtype], p(x), p(w), p(b), p(ga), p(be), p(r), p(out), M, N, K, geglu,
                                                   C.c_float(1e-5), None),
           (M, No), ref, 2 * TOL[dtype], "linear_ln %s M=%d N=%d K=%d" % (dtype, M, N, K), reps=REPS_LN640 if K == 640 else None)


@pytest.mark.parametrize("dtype", ["fp16", "fp32"])
@pytest.mark.parametrize("M,N,K,res", [(4130, 320, 320, 1), (8200, 640, 640, 0),       # streaming kernel, plain
                                       (131072, 320, 1280, 1), (65536, 1280, 256, 0),   # LDS-DMA GEMM, large grids
                                       (115000, 320, 1280, 1),                          # one-shot wide LDS-DMA GEMM (ragged M)
                                       (300, 640, 1280, 1), (512, 1280, 5120, 0)])      # 4-wave tile kernel, split-K
def test_linear_repeat(lib, dtype, M, N, K, res):
    if dtype == "fp32" and M > 100000:
        pytest.skip("covered in fp16; keeps the suite short")
    g = torch.Generator().manual_seed(M + N + K)
    x = q(torch.randn(M, K, generator=g), dtype).cuda()
    w = q(torch.randn(N, K, generator=g) / K ** 0.5, dtype).cuda()
    b = torch.randn(N, generator=g).cuda()
    r = q(torch.randn(M, N, generator=g), dtype).cuda() if res else None
    ref = F.linear(x, w, b)
    ref = (ref + r if res else ref).cpu()
    repeat(lib, lambda out: lib.tango_op_linear(DT[dtype], p(x), p(w), p(b), p(r), p(out), M, N, K, 0, 0, 0, None),
           (M, N), ref, TOL[dtype], "linear %s M=%d N=%d K=%d" % (dtype, M, N, K))


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("M,N,K,res", [(57344, 320, 160, 1), (65536, 640, 2560, 0), (16384, 1280, 1280, 1), (57344, 320, 128, 0),
                                       (4096, 1280, 1280, 1), (8192, 640, 2560, 0),      # split-K x4 / x4 (64 tiles of 256 x 320)
                                       (57344, 320, 96, 1), (57344, 320, 64, 0), (57344, 320, 32, 1)])   # 3, 2, 1 k-chunks (< ring depth)
def test_wide_gemm_repeat(lib, dtype, M, N, K, res):
    """256 x 320 ping-pong LDS-DMA GEMM (gemm_wide.hip): >= 224 tiles, k-chunk counts 4 (the minimum: prologue == whole K),
    5, 40 and 80, with and without the residual epilogue, both 16-bit types; the last two cases run split-K (fp32 partial
    tiles + the ordered reduce kernel)"""
    g = torch.Generator().manual_seed(M + N + K + res)
    x = q(torch.randn(M, K, generator=g), dtype).cuda()
    w = q(torch.randn(N, K, generator=g) / K ** 0.5, dtype).cuda()
    b = torch.randn(N, generator=g).cuda()
    r = q(torch.randn(M, N, generator=g), dtype).cuda() if res else None
    ref = F.linear(x, w, b)
    ref = (ref + r if res else ref).cpu()
    repeat(lib, lambda out: lib.tango_op_linear(DT[dtype], p(x), p(w), p(b), p(r), p(out), M, N, K, 0, 0, 0, None),
           (M, N), ref, TOL[dtype], "wide linear %s M=%d N=%d K=%d" % (dtype, M, N, K))


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("M,N,K,geglu,res,mean", [(57344, 320, 320, 0, 0, 0.7), (32768, 2560, 320, 1, 0, 0.7), (16384, 1280, 1280, 0, 0, -0.3),
                                                  (57344, 320, 640, 0, 1, 6.0)])
def test_wide_gemm_ln_repeat(lib, dtype, M, N, K, geglu, res, mean):
    """folded LayerNorm on the 256 x 320 GEMM (row statistics from the activation fragments in the main loop, per-column
    constants through LDS): plain / GEGLU / residual epilogues, K = 1280 (no streaming-kernel equivalent), rows with |mean| >> std"""
    g = torch.Generator().manual_seed(M + N + K)
    x = q(torch.randn(M, K, generator=g) * (0.5 if mean > 3 else 1.3) + mean, dtype).cuda()
    w = q(torch.randn(N, K, generator=g) / K ** 0.5, dtype).cuda()
    b = torch.randn(N, generator=g).cuda()
    ga, be = (1 + 0.2 * torch.randn(K, generator=g)).cuda(), (0.3 * torch.randn(K, generator=g)).cuda()
    No = N // 2 if geglu else N
    r = q(torch.randn(M, No, generator=g), dtype).cuda() if res else None
    h = F.linear(F.layer_norm(x, (K,), ga, be, 1e-5), w, b)
    if geglu:
        v, gt = h.chunk(2, dim=-1)
        h = v * F.gelu(gt)
    ref = (h + r if res else h).cpu()
    del h
    tol = 2 * TOL[dtype] * (4 if mean > 3 else 1)      # x is stored in T: |mean| / std = 12 amplifies its rounding
    repeat(lib, lambda out: lib.tango_op_linear_ln(DT[dtype], p(x), p(w), p(b), p(ga), p(be), p(r), p(out), M, N, K, geglu,
                                                   C.c_float(1e-5), None),
           (M, No), ref, tol, "wide linear_ln %s M=%d N=%d K=%d" % (dtype, M, N, K))


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("M,N,K,mean", [(16384, 5120, 640, 0.7), (16384, 2560, 1280, -0.3), (65536, 1280, 640, 6.0)])
def test_wide_gemm_ln_xstats_geglu_repeat(lib, dtype, M, N, K, mean):
    """round 4: GEGLU projection with the LayerNorm folded into the weights and the row statistics from a read-only pass
    (ln_stats_kernel + gemm_wide_kernel XS): the shapes of levels 1-2, rows with |mean| >> std"""
    g = torch.Generator().manual_seed(M + N + K)
    x = q(torch.randn(M, K, generator=g) * (0.5 if mean > 3 else 1.3) + mean, dtype).cuda()
    w = q(torch.randn(N, K, generator=g) / K ** 0.5, dtype).cuda()
    b = torch.randn(N, generator=g).cuda()
    ga, be = (1 + 0.2 * torch.randn(K, generator=g)).cuda(), (0.3 * torch.randn(K, generator=g)).cuda()
    h = F.linear(F.layer_norm(x, (K,), ga, be, 1e-5), w, b)
    v, gt = h.chunk(2, dim=-1)
    ref = (v * F.gelu(gt)).cpu()
    del h, v, gt
    tol = 2 * TOL[dtype] * (4 if mean > 3 else 1)
    repeat(lib, lambda out: lib.tango_op_linear_ln(DT[dtype], p(x), p(w), p(b), p(ga), p(be), None, p(out), M, N, K, 1,
                                                   C.c_float(1e-5), None),
           (M, N // 2), ref, tol, "wide linear_ln xstats %s M=%d N=%d K=%d" % (dtype, M, N, K), reps=20)


@pytest.mark.parametrize("dtype", ["fp16", "bf16", "fp32"])
@pytest.mark.parametrize("B,S,Ch,K,ln", [(16, 4096, 320, 320, 1),     # level 0: streaming kernel (K = 320 rows), scalar V^T stores
                                        (64, 1024, 640, 640, 1),     # level 1: wide GEMM, folded LN, LDS-transposed V^T tiles
                                        (64, 256, 1280, 1280, 1),    # level 2: wide GEMM with folded LN at K = 1280
                                        (64, 1024, 640, 640, 0),     # wide GEMM, no LayerNorm
                                        (3, 200, 64, 96, 1)])        # small / ragged: tile kernels, LN kernel + GEMM fallback
def test_linear_qkv_vt_repeat(lib, dtype, B, S, Ch, K, ln):
    """fused q | k | v^T projection (EPI_VT) through every kernel that implements it"""
    if dtype == "fp32" and B * S > 20000:
        pytest.skip("fp32 runs the generic kernels: covered by the small case")
    g = torch.Generator().manual_seed(B + S + Ch + K)
    x = q(torch.randn(B * S, K, generator=g) * 1.2 + 0.4, dtype).cuda()
    w = q(torch.randn(3 * Ch, K, generator=g) / K ** 0.5, dtype).cuda()
    ga, be = (1 + 0.2 * torch.randn(K, generator=g)).cuda(), (0.3 * torch.randn(K, generator=g)).cuda()
    h = F.linear(F.layer_norm(x, (K,), ga, be, 1e-5) if ln else x, w)
    ref_qk = h[:, :2 * Ch].cpu()
    ref_vt = h[:, 2 * Ch:].reshape(B, S, Ch).transpose(1, 2).contiguous().cpu()
    del h
    first = None
    for rep in range(REPS):
        oqk = torch.zeros(B * S, 2 * Ch, device="cuda")
        ovt = torch.zeros(B, Ch, S, device="cuda")
        rc = lib.tango_op_linear_qkv(DT[dtype], p(x), p(w), p(ga) if ln else None, p(be) if ln else None, p(oqk), p(ovt), B, S, Ch, K,
                                     C.c_float(1e-5), None)
        assert rc == 0, lib.tango_last_error().decode()
        if first is None:
            first = (oqk, ovt)
            scale = ref_qk.abs().max().item()
            e1 = ((oqk.cpu() - ref_qk).abs().max() / scale).item()
            e2 = ((ovt.cpu() - ref_vt).abs().max() / scale).item()
            print("qkv %s B=%d S=%d C=%d K=%d ln=%d: rel err qk %.3e vt %.3e" % (dtype, B, S, Ch, K, ln, e1, e2))
            assert e1 <= 2 * TOL[dtype] and e2 <= 2 * TOL[dtype]
        else:
            assert torch.equal(oqk, first[0]) and torch.equal(ovt, first[1]), "repetition %d differs" % rep


@pytest.mark.parametrize("dtype,M,C,K", [("fp16", 32768, 640, 640), ("bf16", 32768, 640, 640), ("fp16", 16384, 1280, 1280), ("fp32", 32768, 320, 512)])
def test_persistent_gemm_geglu_repeat(lib, dtype, M, C, K):
    """LDS-DMA GEMM (gemm_dma.hip) with the fused GEGLU epilogue: x [M, K] @ W [8C, K] -> value * gelu(gate) [M, 4C]
    (>= 512 tiles of 256 x 128; the plain / residual epilogue of that kernel is covered by test_linear_repeat's large cases)"""
    g = torch.Generator().manual_seed(M + C + K)
    x = q(torch.randn(M, K, generator=g), dtype).cuda()
    w = q(torch.randn(8 * C, K, generator=g) / K ** 0.5, dtype).cuda()
    b = torch.randn(8 * C, generator=g).cuda()
    h = F.linear(x, w, b)
    v, gt = h.chunk(2, dim=-1)
    ref = (v * F.gelu(gt)).cpu()
    del h, v, gt
    repeat(lib, lambda out: lib.tango_op_linear(DT[dtype], p(x), p(w), p(b), None, p(out), M, 8 * C, K, 0, 0, 1, None),
           (M, 4 * C), ref, TOL[dtype], "dma GEGLU %s M=%d N=%d K=%d" % (dtype, M, 8 * C, K))


@pytest.mark.parametrize("dtype", ["fp16", "bf16", "fp32"])
@pytest.mark.parametrize("B,Cin,Cout,H,W,ups", [(16, 64, 640, 64, 16, 0), (64, 192, 640, 64, 4, 0), (32, 64, 512, 8, 64, 0),
                                                (256, 64, 512, 8, 8, 0), (16, 64, 640, 32, 8, 1), (4, 128, 320, 16, 16, 0)])
def test_conv3x3_repeat(lib, dtype, B, Cin, Cout, H, W, ups):
    """halo-reuse conv (BN 160 / 128, row tiles, whole-image tiles, fused upsample) and the generic gather kernel"""
    g = torch.Generator().manual_seed(Cin * Cout + H + W)
    x = q(torch.randn(B, Cin, H, W, generator=g), dtype).cuda()
    w = q(torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5, dtype).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    ref = F.conv2d(xin, w, b, padding=1).cpu()
    repeat(lib, lambda out: lib.tango_op_conv2d(DT[dtype], p(x), p(w), p(b), p(out), B, Cin, H, W, Cout, 1, ups, None),
           tuple(ref.shape), ref, TOL[dtype], "conv3x3 %s" % dtype)


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("B,Cin,Cout,H,W,ups", [(64, 64, 320, 64, 16, 0),      # W = 16: 16-row tiles, halo 18 x 18
                                                (64, 96, 320, 128, 8, 0),      # W = 8, three 64-byte channel chunks
                                                (256, 32, 320, 64, 4, 0),      # whole-image tiles (H * W = 256), one chunk
                                                (32, 64, 640, 32, 8, 1),       # fused nearest x2 upsample, two column tiles
                                                (28, 160, 640, 32, 32, 0),     # W = 32: 8-row tiles
                                                (16, 512, 320, 64, 16, 0),     # 64 tiles -> split-K x4 over the 16 channel chunks
                                                (64, 256, 1280, 32, 2, 0)])    # level 3: four 32 x 2 images per tile (halo 544), split-K x2
def test_conv3x3_wide_repeat(lib, dtype, B, Cin, Cout, H, W, ups):
    """halo-reuse conv on the 256 x 320 tile (conv_wide.hip): >= 224 tiles, every tile geometry the UNet uses"""
    g = torch.Generator().manual_seed(Cin * Cout + H + W)
    x = q(torch.randn(B, Cin, H, W, generator=g), dtype).cuda()
    w = q(torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5, dtype).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    ref = F.conv2d(xin, w, b, padding=1).cpu()
    repeat(lib, lambda out: lib.tango_op_conv2d(DT[dtype], p(x), p(w), p(b), p(out), B, Cin, H, W, Cout, 1, ups, None),
           tuple(ref.shape), ref, TOL[dtype], "conv3x3 wide %s" % dtype)


@pytest.mark.parametrize("dtype", ["fp16", "bf16", "fp32"])
@pytest.mark.parametrize("B,heads,Sq,Skv,masked", [(2, 5, 4096, 4096, False), (4, 10, 1024, 1024, False), (4, 5, 4096, 64, True), (3, 2, 200, 7, True)])
def test_attention_repeat(lib, dtype, B, heads, Sq, Skv, masked):
    if dtype == "fp32" and Sq * Skv > 1 << 21:
        pytest.skip("fp32 long-sequence case is covered by the UNet parity tests")
    g = torch.Generator().manual_seed(Sq + Skv)
    C_ = heads * 64
    qq = q(torch.randn(B, Sq, C_, generator=g), dtype).cuda()
    k = q(torch.randn(B, Skv, C_, generator=g), dtype).cuda()
    v = q(torch.randn(B, Skv, C_, generator=g), dtype).cuda()
    bias = None
    if masked:
        m = torch.ones(B, Skv)
        m[0, 1:] = 0
        m[1, Skv // 2:] = 0
        bias = ((1 - m) * -10000.0).cuda()
    qh = qq.view(B, Sq, heads, 64).transpose(1, 2)
    kh = k.view(B, Skv, heads, 64).transpose(1, 2)
    vh = v.view(B, Skv, heads, 64).transpose(1, 2)
    sc = qh @ kh.transpose(-1, -2) * 0.125
    if bias is not None:
        sc = sc + bias[:, None, None, :]
    ref = (sc.softmax(-1) @ vh).transpose(1, 2).reshape(B, Sq, C_).cpu()
    del sc
    repeat(lib, lambda out: lib.tango_op_attention(DT[dtype], p(qq), p(k), p(v), p(bias), p(out), B, heads, Sq, Skv, C.c_float(0.125), None),
           (B, Sq, C_), ref, TOL[dtype], "attention %s" % dtype)


@pytest.mark.parametrize("dtype", ["fp16", "bf16", "fp32"])
@pytest.mark.parametrize("mean,std,outliers", [(50.0, 0.5, False), (-20.0, 2.0, False), (0.0, 1.0, True), (8.0, 0.25, True)])
def test_linear_ln_large_mean_and_outliers(lib, dtype, mean, std, outliers):
    """the folded LayerNorm subtracts mean * wsum AFTER the GEMM and uses statistics of the un-normalised rows.
    Rows with |mean| >> std and rows with outlier channels (what real residual streams look like) must still match
    LayerNorm -> Linear of the same stored inputs.  The tolerance scales with |x|max / std of the row: x is STORED in T,
    so its rounding error relative to the normalised value grows by that factor (inherent to the storage, not the fold)."""
    M, N, K = 4096, 320, 320          # 640-byte rows in fp16 / bf16, 1280-byte rows in fp32: the streaming kernel in all three
    g = torch.Generator().manual_seed(int(abs(mean) * 10 + std * 100) + outliers)
    x = torch.randn(M, K, generator=g) * std + mean
    if outliers:
        idx = torch.randint(0, K, (M, 2), generator=g)
        x.scatter_(1, idx, 40.0 * std * torch.sign(torch.randn(M, 2, generator=g)))
    x = q(x, dtype).cuda()
    w = q(torch.randn(N, K, generator=g) / K ** 0.5, dtype).cuda()
    b = torch.randn(N, generator=g).cuda()
    ga, be = (1 + 0.2 * torch.randn(K, generator=g)).cuda(), (0.3 * torch.randn(K, generator=g)).cuda()
    ref = F.linear(F.layer_norm(x.double(), (K,), ga.double(), be.double(), 1e-5), w.double(), b.double()).float().cpu()
    out = torch.zeros(M, N, device="cuda")
    rc = lib.tango_op_linear_ln(DT[dtype], p(x), p(w), p(b), p(ga), p(be), None, p(out), M, N, K, 0, C.c_float(1e-5), None)
    assert rc == 0, lib.tango_last_error().decode()
    err = ((out.cpu() - ref).abs().max() / ref.abs().max()).item()
    rowstd = x.std(dim=1).mean().item()
    # error model: W' = round_T(W * gamma) multiplies only (x - mean) because wsum is summed from the ROUNDED W' (the
    # mean * wsum term cancels exactly), so the storage type contributes the usual op tolerance; what grows with
    # |x|max / std is the fp32 accumulation of sum_k W'x (terms of size |x|max) and the one-pass variance: ~40 ulp(fp32)
    tol = 2 * TOL[dtype] + 40 * (x.abs().max().item() / (rowstd + 1e-9)) * 6e-8
    print("linear_ln %s mean %.1f std %.2f outliers %s: rel err %.3e (tol %.3e)" % (dtype, mean, std, outliers, err, tol))
    assert err <= tol
