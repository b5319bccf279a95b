"""gemm_duo.hip (256 x 160 tile, four waves, two workgroups per CU): every epilogue it instantiates, against the fp32 torch
statement of the op, repeated bit-identically, and -- where the 256 x 320 kernel takes the same problem -- bit-equal to that
kernel (same wave tile, same k order, same epilogue arithmetic: the two must agree to the last bit).

The dispatch switches are launch-time (tuning.h): the tests set the environment, call tango_tuning_reload(), and restore it."""
import contextlib
import ctypes as C
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = {"fp16": 1, "bf16": 2}
TOL = {"fp16": 4e-3, "bf16": 3e-2}
REPS = int(os.environ.get("TANGO_DUO_REPS", "12"))


def q(t, dtype):
    return t.half().float() if dtype == "fp16" else t.bfloat16().float()


def p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


@contextlib.contextmanager
def tuning(lib, **env):
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    lib.tango_tuning_reload()
    try:
        yield
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        lib.tango_tuning_reload()


DUO = dict(TANGO_DUO_MAXK=100000, TANGO_DUO_MIN_TILES=1, TANGO_DUO_MASK=15)
NODUO = dict(TANGO_DUO_MAXK=0)


def run(lib, call, shape, reps=1):
    first = None
    for rep in range(reps):
        out = torch.zeros(shape, device="cuda")
        rc = call(out)
        assert rc == 0, lib.tango_last_error().decode()
        if first is None:
            first = out
        else:
            assert torch.equal(out, first), "repetition %d differs at %d elements" % (rep, (out != first).sum().item())
    return first


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("M,N,K,res,geglu", [
    (512, 320, 320, 1, 0),          # four tiles: the arithmetic without any co-residency
    (65536, 640, 640, 1, 0),        # 1024 tiles, 20 chunks, residual (levels' to_out / proj_out shape)
    (57344, 320, 1280, 1, 0),       # FF out at level 0
    (16384, 2560, 640, 0, 1),       # GEGLU epilogue
    (16384, 1280, 2560, 0, 0),      # 80 chunks
    (57344, 320, 32, 1, 0), (57344, 320, 64, 0, 0), (57344, 480, 96, 1, 0),     # 1, 2, 3 chunks (<= ring depth); N = 3 x 160
])
def test_duo_linear(lib, dtype, M, N, K, res, geglu):
    g = torch.Generator().manual_seed(M + N + K + res)
    x = q(torch.randn(M, K, generator=g), dtype).cuda()
    w = q(torch.randn(N, K, generator=g) / K ** 0.5, dtype).cuda()
    b = torch.randn(N, generator=g).cuda()
    No = N // 2 if geglu else N
    r = q(torch.randn(M, No, generator=g), dtype).cuda() if res else None
    h = F.linear(x, w, b)
    if geglu:
        v, gt = h.chunk(2, dim=-1)
        h = v * F.gelu(gt)
    ref = h + r if res else h
    del h
    call = lambda out: lib.tango_op_linear(DT[dtype], p(x), p(w), p(b), p(r), p(out), M, N, K, 0, 0, geglu, None)
    with tuning(lib, **DUO):
        out = run(lib, call, (M, No), REPS)
    err = ((out - ref).abs().max() / (ref.abs().max() + 1e-9)).item()
    assert err <= TOL[dtype], "duo linear %s M=%d N=%d K=%d: rel err %.3e" % (dtype, M, N, K, err)
    if N % 320 == 0 and (M // 256) * (N // 320) >= 192 and K >= 32:
        with tuning(lib, **NODUO):
            wide = run(lib, call, (M, No))
        assert torch.equal(out, wide), "duo vs 256x320 kernel: %d elements differ" % (out != wide).sum().item()


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("M,N,K,geglu,res,mean", [(57344, 320, 320, 0, 0, 0.7), (57344, 960, 320, 0, 1, 0.7), (16384, 1280, 1280, 0, 0, -0.3),
                                                  (32768, 2560, 320, 1, 0, 0.7), (57344, 320, 640, 0, 1, 6.0)])
def test_duo_linear_ln(lib, dtype, M, N, K, geglu, res, mean):
    """folded LayerNorm (row statistics from the activation fragments in the main loop), plain / residual / GEGLU epilogues"""
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
    ref = h + r if res else h
    del h
    call = lambda out: lib.tango_op_linear_ln(DT[dtype], p(x), p(w), p(b), p(ga), p(be), p(r), p(out), M, N, K, geglu, C.c_float(1e-5), None)
    with tuning(lib, **DUO):
        out = run(lib, call, (M, No), REPS)
    tol = 2 * TOL[dtype] * (4 if mean > 3 else 1)
    err = ((out - ref).abs().max() / (ref.abs().max() + 1e-9)).item()
    assert err <= tol, "duo linear_ln %s M=%d N=%d K=%d: rel err %.3e" % (dtype, M, N, K, err)


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("B,S,Ch,K,ln", [(64, 1024, 640, 640, 1), (64, 256, 1280, 1280, 1), (64, 1024, 640, 640, 0), (14, 4096, 320, 320, 1)])
def test_duo_linear_qkv_vt(lib, dtype, B, S, Ch, K, ln):
    """fused q | k | v^T projection (EPI_VT: the V columns leave through the LDS transpose), with and without folded LayerNorm"""
    g = torch.Generator().manual_seed(B + S + Ch + K)
    x = q(torch.randn(B * S, K, generator=g) * 1.2 + 0.4, dtype).cuda()
    w = q(torch.randn(3 * Ch, K, generator=g) / K ** 0.5, dtype).cuda()
    ga, be = (1 + 0.2 * torch.randn(K, generator=g)).cuda(), (0.3 * torch.randn(K, generator=g)).cuda()
    h = F.linear(F.layer_norm(x, (K,), ga, be, 1e-5) if ln else x, w)
    ref_qk = h[:, :2 * Ch].clone()
    ref_vt = h[:, 2 * Ch:].reshape(B, S, Ch).transpose(1, 2).contiguous()
    del h

    def once():
        oqk = torch.zeros(B * S, 2 * Ch, device="cuda")
        ovt = torch.zeros(B, Ch, S, device="cuda")
        rc = lib.tango_op_linear_qkv(DT[dtype], p(x), p(w), p(ga) if ln else None, p(be) if ln else None, p(oqk), p(ovt), B, S, Ch, K,
                                     C.c_float(1e-5), None)
        assert rc == 0, lib.tango_last_error().decode()
        return oqk, ovt

    with tuning(lib, **DUO):
        first = once()
        for rep in range(1, REPS):
            o = once()
            assert torch.equal(o[0], first[0]) and torch.equal(o[1], first[1]), "repetition %d differs" % rep
    scale = ref_qk.abs().max().item()
    e1 = ((first[0] - ref_qk).abs().max() / scale).item()
    e2 = ((first[1] - ref_vt).abs().max() / scale).item()
    assert e1 <= 2 * TOL[dtype] and e2 <= 2 * TOL[dtype], "duo qkv %s: rel err qk %.3e vt %.3e" % (dtype, e1, e2)
    if K >= 640:      # the 256 x 320 kernel takes these too (K = 320 rows go to the streaming kernel at this size)
        with tuning(lib, **NODUO):
            wide = once()
        assert torch.equal(first[0], wide[0]) and torch.equal(first[1], wide[1]), "duo vs 256x320 kernel differ"


def test_duo_routing_labels(lib):
    """the dispatch switch reaches the engine: with the kernel enabled, a UNet step's short-K linears carry the (duo) label"""
    from tango_amd.engine import UNET_CONFIG_LARGE, Engine
    with tuning(lib, TANGO_DUO_MAXK=1280, TANGO_DUO_MIN_TILES=384, TANGO_DUO_MASK=7):
        e = Engine(unet=UNET_CONFIG_LARGE, dtype="fp16")
        e.load_synthetic(1)
        rows = e.profile_unet(16, 64)
    labs = {r[0].split(" ")[0] for r in rows}
    assert "linear(duo)" in labs, sorted(labs)


# ---- persistent form of the 256 x 320 GEMM (gemm_wide_pers_kernel): one workgroup per CU walks its tiles, the next tile's first chunk is
# requested behind the epilogue.  Same arithmetic in the same order: bit-equal to the one-tile-per-workgroup kernel. ----

PERS = dict(TANGO_WIDE_PERS=1, TANGO_DUO_MAXK=0)
NOPERS = dict(TANGO_WIDE_PERS=0, TANGO_DUO_MAXK=0)


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("M,N,K,res,geglu", [
    (65536, 640, 640, 1, 0),        # 512 tiles: two per workgroup, residual epilogue
    (19200, 2560, 640, 0, 1),       # 600 tiles: 88 workgroups walk three tiles, the rest two; GEGLU epilogue
    (131072, 320, 64, 1, 0),        # two k-chunks (< ring depth) per tile
    (131072, 320, 32, 0, 0),        # one
    (262144, 320, 320, 1, 0),       # four tiles per workgroup (the level-0 linears)
])
def test_wide_pers_linear_bit_equal(lib, dtype, M, N, K, res, geglu):
    g = torch.Generator().manual_seed(M + N + K + res)
    x = q(torch.randn(M, K, generator=g), dtype).cuda()
    w = q(torch.randn(N, K, generator=g) / K ** 0.5, dtype).cuda()
    b = torch.randn(N, generator=g).cuda()
    No = N // 2 if geglu else N
    r = q(torch.randn(M, No, generator=g), dtype).cuda() if res else None
    call = lambda out: lib.tango_op_linear(DT[dtype], p(x), p(w), p(b), p(r), p(out), M, N, K, 0, 0, geglu, None)
    with tuning(lib, **NOPERS):
        ref = run(lib, call, (M, No))
    with tuning(lib, **PERS):
        out = run(lib, call, (M, No), REPS)
    assert torch.equal(out, ref), "persistent vs one-shot 256x320 kernel: %d elements differ" % (out != ref).sum().item()
    h = F.linear(x, w, b)
    if geglu:
        v, gt = h.chunk(2, dim=-1)
        h = v * F.gelu(gt)
    h = h + r if res else h
    err = ((out - h).abs().max() / (h.abs().max() + 1e-9)).item()
    assert err <= TOL[dtype], err


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("M,N,K,geglu,res", [(131072, 960, 320, 0, 1), (65536, 1920, 640, 0, 0), (32768, 5120, 640, 1, 0), (16384, 10240, 1280, 1, 0)])
def test_wide_pers_linear_ln_bit_equal(lib, dtype, M, N, K, geglu, res):
    """folded LayerNorm: in-loop statistics (plain / residual epilogue) and the external-statistics GEGLU form (levels 1-2)"""
    g = torch.Generator().manual_seed(M + N + K)
    x = q(torch.randn(M, K, generator=g) * 1.3 + 0.7, dtype).cuda()
    w = q(torch.randn(N, K, generator=g) / K ** 0.5, dtype).cuda()
    b = torch.randn(N, generator=g).cuda()
    ga, be = (1 + 0.2 * torch.randn(K, generator=g)).cuda(), (0.3 * torch.randn(K, generator=g)).cuda()
    No = N // 2 if geglu else N
    r = q(torch.randn(M, No, generator=g), dtype).cuda() if res else None
    call = lambda out: lib.tango_op_linear_ln(DT[dtype], p(x), p(w), p(b), p(ga), p(be), p(r), p(out), M, N, K, geglu, C.c_float(1e-5), None)
    with tuning(lib, **NOPERS):
        ref = run(lib, call, (M, No))
    with tuning(lib, **PERS):
        out = run(lib, call, (M, No), REPS)
    assert torch.equal(out, ref), "persistent vs one-shot (LN): %d elements differ" % (out != ref).sum().item()


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("B,S,Ch,K,ln", [(64, 1024, 640, 640, 1), (64, 1024, 640, 640, 0), (128, 256, 1280, 1280, 1)])
def test_wide_pers_qkv_vt_bit_equal(lib, dtype, B, S, Ch, K, ln):
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

    with tuning(lib, **NOPERS):
        ref = once()
    with tuning(lib, **PERS):
        for rep in range(REPS):
            o = once()
            assert torch.equal(o[0], ref[0]) and torch.equal(o[1], ref[1]), "repetition %d differs from the one-shot kernel" % rep
