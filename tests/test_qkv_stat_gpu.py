"""qkv_stat_kernel (ff_fused.hip): norm1 + [to_q | to_k | to_v^T] of the level-0 self-attention on the activation-stationary kernel
(reference: mustango/diffusers/src/diffusers/models/attention.py:276-296, attention_processor.py:495-520) against the fp64 torch statement
of the op, against the GEMM route, repeated bit-identically, timed at the benchmarked size."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from test_duo_gpu import DT, p, q, tuning

pytestmark = pytest.mark.gpu
TOL = {"fp16": 4e-3, "bf16": 3e-2}
Ch = 320


def make(B, S, dtype, seed, mean):
    g = torch.Generator().manual_seed(seed)
    x = q(torch.randn(B * S, Ch, generator=g) * 1.2 + mean, dtype).cuda()
    w = q(torch.randn(3 * Ch, Ch, generator=g) / Ch ** 0.5, dtype).cuda()
    ga, be = (1 + 0.2 * torch.randn(Ch, generator=g)).cuda(), (0.3 * torch.randn(Ch, generator=g)).cuda()
    return x, w, ga, be


def call(lib, dtype, t, B, S, mode, reps=0):
    x, w, ga, be = t
    qk = torch.zeros(B * S, 2 * Ch, device="cuda")
    vt = torch.zeros(B, Ch, S, device="cuda")
    ms = C.c_float(0.0)
    rc = lib.tango_op_qkv_stat(DT[dtype], p(x), p(w), p(ga), p(be), p(qk), p(vt), B, S, Ch, C.c_float(1e-5), mode, reps, C.byref(ms) if reps else None, None)
    assert rc == 0, lib.tango_last_error().decode()
    return qk, vt, ms.value


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("B,S,mean", [(1, 256, 0.4), (3, 1024, -0.6), (14, 4096, 2.0)])
def test_qkv_stat_matches_torch_and_gemm_route(lib, dtype, B, S, mean):
    t = make(B, S, dtype, B * S + 1, mean)
    x, w, ga, be = t
    h = F.linear(F.layer_norm(x.double(), (Ch,), ga.double(), be.double(), 1e-5), w.double())
    ref_qk = h[:, :2 * Ch].float()
    ref_vt = h[:, 2 * Ch:].reshape(B, S, Ch).transpose(1, 2).contiguous().float()
    qk, vt, _ = call(lib, dtype, t, B, S, 0)
    sc = h.abs().max().item()
    e_qk = (qk - ref_qk).abs().max().item() / sc
    e_vt = (vt - ref_vt).abs().max().item() / sc
    msg = "qkv_stat %s B=%d S=%d: q|k vs fp64 %.3e, v^T vs fp64 %.3e" % (dtype, B, S, e_qk, e_vt)
    if B * S >= 4096:
        qk2, vt2, _ = call(lib, dtype, t, B, S, 1)
        d = max((qk - qk2).abs().max().item(), (vt - vt2).abs().max().item()) / sc
        e2 = max((qk2 - ref_qk).abs().max().item(), (vt2 - ref_vt).abs().max().item()) / sc
        msg += ", GEMM route vs fp64 %.3e, the two routes apart %.3e" % (e2, d)
        assert d <= TOL[dtype]
    print(msg)
    assert e_qk <= TOL[dtype] and e_vt <= TOL[dtype]
    for rep in range(8):
        qk3, vt3, _ = call(lib, dtype, t, B, S, 0)
        assert torch.equal(qk3, qk) and torch.equal(vt3, vt), "repetition %d differs" % rep


def _qkv_c(lib, dtype, B, S, Cc, how, perm, seed):
    """how: "stat" = qkv_stat_kernel, "ln" / "plain" = the engine's GEMM dispatch with / without the LayerNorm (tango_op_linear_qkv_perm)"""
    g = torch.Generator().manual_seed(seed)
    x = q(torch.randn(B * S, Cc, generator=g) * 1.2 + 0.3, dtype).cuda()
    w = q(torch.randn(3 * Cc, Cc, generator=g) / Cc ** 0.5, dtype).cuda()
    ga, be = (1 + 0.2 * torch.randn(Cc, generator=g)).cuda(), (0.3 * torch.randn(Cc, generator=g)).cuda()
    qk = torch.zeros(B * S, 2 * Cc, device="cuda")
    vt = torch.zeros(B, Cc, S, device="cuda")
    if how == "stat":
        rc = lib.tango_op_qkv_stat(DT[dtype], p(x), p(w), p(ga), p(be), p(qk), p(vt), B, S, Cc, C.c_float(1e-5), 2 * perm, 0, None, None)
    else:
        ln = how == "ln"
        rc = lib.tango_op_linear_qkv_perm(DT[dtype], p(x), p(w), p(ga) if ln else None, p(be) if ln else None, p(qk), p(vt), B, S, Cc, Cc, C.c_float(1e-5), perm, None)
    assert rc == 0, lib.tango_last_error().decode()
    return qk, vt


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("Cc,B,S", [(320, 1, 256), (320, 2, 1024), (320, 16, 4096), (640, 1, 256), (640, 3, 1024), (640, 64, 1024), (1280, 2, 256), (1280, 64, 256)])
def test_vt_in_fragment_order_from_every_producer(lib, dtype, Cc, B, S):
    """GemmParams / QKVParams::vt_perm: every producer of v^T (the activation-stationary kernel, and the transposed epilogues of the 256 x 320 / 256 x 160
    GEMMs, the streaming linear and the tile kernels -- whichever the dispatcher picks for the shape, LayerNorm folded or not) writes the SAME values with
    the tokens of every block of 32 in the attention kernel's fragment order: position 8 g + 4 hi + r holds token 16 hi + 4 g + r (attention.hip VDMA
    reads that by LDS-DMA; reference op: attention_processor.py:495-520, the v projection)"""
    s = torch.arange(S)
    pos = ((s & ~31) | ((s & 12) << 1) | ((s & 16) >> 2) | (s & 3)).cuda()
    assert sorted(pos.tolist()) == list(range(S))
    hows = ["ln", "plain"] + (["stat"] if Cc == 320 and (B * S) % 256 == 0 else [])
    for how in hows:
        qk0, vt0 = _qkv_c(lib, dtype, B, S, Cc, how, 0, 5 + Cc + B)
        qk1, vt1 = _qkv_c(lib, dtype, B, S, Cc, how, 1, 5 + Cc + B)
        assert vt0.abs().max().item() > 0.1
        assert torch.equal(qk0, qk1)
        assert torch.equal(vt1.index_select(2, pos), vt0), "%s: %d elements differ" % (how, (vt1.index_select(2, pos) != vt0).sum().item())


def test_vt_perm_argument_errors(lib):
    x = torch.zeros(48, 320, device="cuda"); w = torch.zeros(960, 320, device="cuda")
    qk = torch.zeros(48, 640, device="cuda"); vt = torch.zeros(1, 320, 48, device="cuda")
    assert lib.tango_op_linear_qkv_perm(DT["fp16"], p(x), p(w), None, None, p(qk), p(vt), 1, 48, 320, 320, C.c_float(1e-5), 1, None) != 0   # S % 32 != 0
    assert b"vt_perm" in lib.tango_last_error()


def test_qkv_stat_time_at_config3_size(lib):
    B, S = 64, 4096
    t = make(B, S, "fp16", 9, 0.3)
    ms = {0: [], 1: []}
    for rnd in range(3):
        for mode in (1, 0):
            ms[mode].append(call(lib, "fp16", t, B, S, mode, reps=20)[2])
    a, b = sorted(ms[0])[1], sorted(ms[1])[1]
    gf = 2.0 * B * S * 3 * Ch * Ch / 1e9
    print("qkv M=%d fp16: activation-stationary kernel %.3f ms (%.0f TFLOP/s), GEMM route %.3f ms (%.0f TFLOP/s)" % (B * S, a, gf / a, b, gf / b))
    assert a <= b * 1.05


@pytest.mark.parametrize("dtype,tol", [("fp16", 4e-3), ("bf16", 4e-2)])
def test_unet_forward_with_qkv_stat(lib, dtype, tol):
    from oracle import tango_oracle as O
    from tango_amd import weights as W
    from tango_amd.engine import UNET_CONFIG_LARGE, Engine
    B2 = 16                                              # 65536 rows at level 0: the engine's threshold
    g = torch.Generator().manual_seed(35)
    x = torch.randn(B2, 8, 256, 16, generator=g)
    enc = torch.randn(B2, 64, 1024, generator=g)
    mask = torch.ones(B2, 64, dtype=torch.bool)
    mask[: B2 // 2, 1:] = False
    e = Engine(unet=UNET_CONFIG_LARGE, dtype=dtype)
    e.load_synthetic(1234)
    outs = {}
    for on in (0, 1):
        with tuning(lib, TANGO_QKV_STAT=on):
            e.drop_plans()
            outs[on] = e.unet_forward(x.cuda(), 500, enc.cuda(), mask.cuda()).cpu()
            labels = [r[0] for r in e.profile_unet(B2, 64)]
        assert any(l.startswith("qkv_stat") for l in labels) == (on == 1), labels[:12]
    e.drop_plans()
    sd = W.synth_state_dict(W.unet_param_shapes(O.UNET_CONFIG_LARGE, "unet."), 1234)
    rows = [1, 14]
    with torch.no_grad():
        ref = O.unet_forward(sd, O.UNET_CONFIG_LARGE, x[rows], 500, enc[rows], mask[rows], prefix="unet.")
    scale = ref.abs().max().item()
    e0 = (outs[0][rows] - ref).abs().max().item() / scale
    e1 = (outs[1][rows] - ref).abs().max().item() / scale
    d = (outs[0] - outs[1]).abs().max().item() / scale
    print("UNet forward %s B2=%d: GEMM-route QKV vs oracle %.3e, activation-stationary QKV vs oracle %.3e, apart %.3e" % (dtype, B2, e0, e1, d))
    assert e0 <= tol and e1 <= tol and d <= tol


@pytest.mark.parametrize("dtype,tol", [("fp16", 4e-3), ("bf16", 4e-2)])
def test_unet_forward_with_groupnorm_proj_in_on_the_stationary_kernel(lib, dtype, tol):
    """Transformer2DModel.norm -> proj_in at level 0 (transformer_2d.py:255-262): GroupNorm statistics pass + the activation-stationary kernel
    normalising the rows it loads (norm mode 2), against the GroupNorm kernel + GEMM route and the oracle; repeated bit-identically"""
    from oracle import tango_oracle as O
    from tango_amd import weights as W
    from tango_amd.engine import UNET_CONFIG_LARGE, Engine
    B2 = 16                                              # 65536 rows at level 0: the engine's threshold
    g = torch.Generator().manual_seed(36)
    x = torch.randn(B2, 8, 256, 16, generator=g)
    x[:, 2] += 2.0                                       # group means away from zero
    enc = torch.randn(B2, 64, 1024, generator=g)
    mask = torch.ones(B2, 64, dtype=torch.bool)
    e = Engine(unet=UNET_CONFIG_LARGE, dtype=dtype)
    e.load_synthetic(1234)
    outs = {}
    for on in (0, 1):
        with tuning(lib, TANGO_GN_PROJ_STAT=on):
            e.drop_plans()
            outs[on] = e.unet_forward(x.cuda(), 500, enc.cuda(), mask.cuda()).cpu()
            if on:
                again = e.unet_forward(x.cuda(), 500, enc.cuda(), mask.cuda()).cpu()
                assert torch.equal(again, outs[1])
            labels = [r[0] for r in e.profile_unet(B2, 64)]
        assert any(l.startswith("linear+gn(stat)") for l in labels) == (on == 1), labels[:12]
    e.drop_plans()
    sd = W.synth_state_dict(W.unet_param_shapes(O.UNET_CONFIG_LARGE, "unet."), 1234)
    rows = [2, 13]
    with torch.no_grad():
        ref = O.unet_forward(sd, O.UNET_CONFIG_LARGE, x[rows], 500, enc[rows], mask[rows], prefix="unet.")
    scale = ref.abs().max().item()
    e0 = (outs[0][rows] - ref).abs().max().item() / scale
    e1 = (outs[1][rows] - ref).abs().max().item() / scale
    d = (outs[0] - outs[1]).abs().max().item() / scale
    print("UNet forward %s B2=%d: GroupNorm kernel + proj_in GEMM vs oracle %.3e, statistics + stationary kernel vs oracle %.3e, apart %.3e" % (dtype, B2, e0, e1, d))
    assert e0 <= tol and e1 <= tol and d <= tol


@pytest.mark.parametrize("dtype,tol", [("fp16", 4e-3), ("bf16", 4e-2)])
def test_unet_forward_with_kv_tiles_by_lds_dma(lib, dtype, tol):
    """self-attention (Sq > 512: levels 0-1) with the K tile, and -- v^T written in fragment order by its producer (qkv_stat_kernel at level 0, the GEMMs'
    transposed epilogues at level 1: AttnParams / GemmParams::vt_perm) -- the V^T tile fetched by
    LDS-DMA (attention.hip KDMA / VDMA; reference op attention_processor.py:495-540): same arithmetic as the register-staged kernel, so the three forms
    must agree BITWISE through a whole UNet forward; and against the oracle"""
    from oracle import tango_oracle as O
    from tango_amd import weights as W
    from tango_amd.engine import UNET_CONFIG_LARGE, Engine
    B2 = 16
    g = torch.Generator().manual_seed(37)
    x = torch.randn(B2, 8, 256, 16, generator=g)
    enc = torch.randn(B2, 64, 1024, generator=g)
    mask = torch.ones(B2, 64, dtype=torch.bool)
    mask[: B2 // 2, 1:] = False
    e = Engine(unet=UNET_CONFIG_LARGE, dtype=dtype)
    e.load_synthetic(1234)
    outs = {}
    for kd, vd in ((0, 0), (1, 0), (1, 1), (1, 2)):
        with tuning(lib, TANGO_ATTN_KDMA=kd, TANGO_ATTN_VDMA=vd):
            e.drop_plans()
            outs[(kd, vd)] = e.unet_forward(x.cuda(), 500, enc.cuda(), mask.cuda()).cpu()
    e.drop_plans()
    assert torch.equal(outs[(0, 0)], outs[(1, 0)]), "K by LDS-DMA: %d elements differ" % (outs[(0, 0)] != outs[(1, 0)]).sum().item()
    assert torch.equal(outs[(0, 0)], outs[(1, 1)]), "K and V^T by LDS-DMA: %d elements differ" % (outs[(0, 0)] != outs[(1, 1)]).sum().item()
    # TANGO_ATTN_VDMA=2 (the default): the GEMM routes' transposed epilogues write the permuted v^T as well (level 1, Sq = 1024)
    assert torch.equal(outs[(0, 0)], outs[(1, 2)]), "V^T in fragment order from the GEMM epilogues: %d elements differ" % (outs[(0, 0)] != outs[(1, 2)]).sum().item()
    sd = W.synth_state_dict(W.unet_param_shapes(O.UNET_CONFIG_LARGE, "unet."), 1234)
    rows = [3, 12]
    with torch.no_grad():
        ref = O.unet_forward(sd, O.UNET_CONFIG_LARGE, x[rows], 500, enc[rows], mask[rows], prefix="unet.")
    err = (outs[(1, 2)][rows] - ref).abs().max().item() / ref.abs().max().item()
    print("UNet forward %s B2=%d with K / V^T tiles by LDS-DMA vs oracle %.3e (bit-identical to the register-staged kernel)" % (dtype, B2, err))
    assert err <= tol
