"""ff_fused.hip: the level-0 feed-forward of BasicTransformerBlock in one launch --
out = x + ff.net.2(GEGLU(ff.net.0.proj(LayerNorm3(x)))) (reference: mustango/diffusers/src/diffusers/models/attention.py:326-335, :338-387,
:412-433) -- against the fp32 torch statement of the op on the rounded operands, against the engine's two-GEMM route, repeated
bit-identically, and inside a whole UNet forward against the oracle.  Tolerances are relative to max |out|."""
import ctypes as C
import os

import pytest
import torch
import torch.nn.functional as F

from test_duo_gpu import DT, p, q, tuning

pytestmark = pytest.mark.gpu

TOL = {"fp16": 4e-3, "bf16": 3e-2}
Cc, H = 320, 1280


def make(M, dtype, seed, mean=0.4, outlier=False):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, Cc, generator=g) * 1.3 + mean
    if outlier:
        x[:, 7] += 9.0            # a hot channel, as the residual stream carries
        x[::5] *= 3.0
    x = q(x, dtype).cuda()
    w1 = q(torch.randn(2 * H, Cc, generator=g) / Cc ** 0.5, dtype).cuda()
    b1 = (0.3 * torch.randn(2 * H, generator=g)).cuda()
    w2 = q(torch.randn(Cc, H, generator=g) / H ** 0.5, dtype).cuda()
    b2 = (0.3 * torch.randn(Cc, generator=g)).cuda()
    ga, be = (1 + 0.2 * torch.randn(Cc, generator=g)).cuda(), (0.3 * torch.randn(Cc, generator=g)).cuda()
    return x, w1, b1, ga, be, w2, b2


def reference(x, w1, b1, ga, be, w2, b2):
    h = F.linear(F.layer_norm(x.double(), (Cc,), ga.double(), be.double(), 1e-5), w1.double(), b1.double())
    v, gt = h.chunk(2, dim=-1)
    return (x.double() + F.linear(v * F.gelu(gt), w2.double(), b2.double())).float()


def call(lib, dtype, t, M, mode, reps=0):
    x, w1, b1, ga, be, w2, b2 = t
    out = torch.zeros(M, Cc, device="cuda")
    ms = C.c_float(0.0)
    rc = lib.tango_op_ff_fused(DT[dtype], p(x), p(w1), p(b1), p(ga), p(be), p(w2), p(b2), p(out), M, Cc, H, C.c_float(1e-5), mode, reps,
                               C.byref(ms) if reps else None, None)
    assert rc == 0, lib.tango_last_error().decode()
    return out, ms.value


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("M,mean,outlier", [(128, 0.4, False), (1024, -0.7, False), (8192, 0.4, True), (57344, 2.5, False)])
def test_ff_fused_matches_torch_and_two_gemm_route(lib, dtype, M, mean, outlier):
    t = make(M, dtype, M + int(mean * 10), mean, outlier)
    ref = reference(*t)
    scale = ref.abs().max().item()
    fused, _ = call(lib, dtype, t, M, 0)
    e_f = (fused - ref).abs().max().item() / scale
    two = None
    if M >= 4096:                # the streaming GEGLU kernel of the two-GEMM route wants >= 4096 rows
        two, _ = call(lib, dtype, t, M, 1)
        e_t = (two - ref).abs().max().item() / scale
        d = (fused - two).abs().max().item() / scale
        print("ff %s M=%d: fused vs fp64 %.3e, two-GEMM route vs fp64 %.3e, fused vs two-GEMM %.3e" % (dtype, M, e_f, e_t, d))
        assert e_t <= TOL[dtype] and d <= TOL[dtype]
    else:
        print("ff %s M=%d: fused vs fp64 %.3e" % (dtype, M, e_f))
    assert e_f <= TOL[dtype]
    for rep in range(10):
        again, _ = call(lib, dtype, t, M, 0)
        assert torch.equal(again, fused), "repetition %d differs at %d elements" % (rep, (again != fused).sum().item())


def test_ff_fused_main_loop_forms_agree_bitwise(lib):
    """the asm-pipelined LDS stream + staged epilogue (shipped) and the compiler-scheduled loop + accumulator-layout epilogue (TANGO_FF_FUSED=2)
    perform the same arithmetic in the same order per output element"""
    M = 16384
    t = make(M, "fp16", 77, 0.9, True)
    a, _ = call(lib, "fp16", t, M, 0)
    with tuning(lib, TANGO_FF_FUSED=2):
        b, _ = call(lib, "fp16", t, M, 0)
    assert torch.equal(a, b), "%d elements differ" % (a != b).sum().item()


def test_ff_fused_refuses_other_shapes(lib):
    t = make(128, "fp16", 1)
    x, w1, b1, ga, be, w2, b2 = t
    out = torch.zeros(100, Cc, device="cuda")
    rc = lib.tango_op_ff_fused(1, p(x), p(w1), p(b1), p(ga), p(be), p(w2), p(b2), p(out), 100, Cc, H, C.c_float(1e-5), 0, 0, None, None)
    assert rc != 0 and b"ff_fused" in lib.tango_last_error()
    rc = lib.tango_op_ff_fused(0, p(x), p(w1), p(b1), p(ga), p(be), p(w2), p(b2), p(out), 128, Cc, H, C.c_float(1e-5), 0, 0, None, None)
    assert rc != 0


@pytest.mark.parametrize("dtype", ["fp16"])
def test_ff_fused_time_at_config3_size(lib, dtype):
    """same-process A/B at the benchmarked size (M = 64 x 4096): printed, and the fused launch must not be slower than the two GEMMs"""
    M = 262144
    t = make(M, dtype, 5)
    ms = {}
    for rnd in range(3):
        for mode in (1, 0, 2):
            with tuning(lib, TANGO_FF_FUSED=2 if mode == 2 else 1):       # 2: the compiler-scheduled main loop of the fused kernel
                _, v = call(lib, dtype, t, M, 0 if mode == 2 else mode, reps=20)
            ms.setdefault(mode, []).append(v)
    f, two, plain = sorted(ms[0])[1], sorted(ms[1])[1], sorted(ms[2])[1]
    gf = 2.0 * M * (2 * H * Cc + H * Cc) / 1e9
    print("ff M=%d %s: fused %.3f ms (%.0f TFLOP/s), fused with the compiler-scheduled loop %.3f ms, two GEMMs %.3f ms (%.0f TFLOP/s)"
          % (M, dtype, f, gf / f, plain, two, gf / two))
    if os.environ.get("TANGO_FF_FUSED", "1") != "0":
        assert f <= two * 1.05


@pytest.mark.parametrize("dtype,tol", [("fp16", 4e-3), ("bf16", 4e-2)])
def test_unet_forward_with_fused_ff(lib, dtype, tol):
    from oracle import tango_oracle as O
    from tango_amd import weights as W
    from tango_amd.engine import UNET_CONFIG_LARGE, Engine
    B2 = 8                                               # 32768 rows at level 0: the engine's threshold for the fused launch
    g = torch.Generator().manual_seed(33)
    x = torch.randn(B2, 8, 256, 16, generator=g)
    enc = torch.randn(B2, 64, 1024, generator=g)
    mask = torch.ones(B2, 64, dtype=torch.bool)
    mask[: B2 // 2, 1:] = False
    e = Engine(unet=UNET_CONFIG_LARGE, dtype=dtype)
    e.load_synthetic(1234)
    outs = {}
    for fused in (0, 1):
        with tuning(lib, TANGO_FF_FUSED=fused):
            e.drop_plans()
            outs[fused] = e.unet_forward(x.cuda(), 500, enc.cuda(), mask.cuda()).cpu()
            labels = [r[0] for r in e.profile_unet(B2, 64)]
        assert any(l.startswith("ff_fused") for l in labels) == (fused == 1), labels[:12]
    e.drop_plans()
    sd = W.synth_state_dict(W.unet_param_shapes(O.UNET_CONFIG_LARGE, "unet."), 1234)
    rows = [0, 5]
    with torch.no_grad():
        ref = O.unet_forward(sd, O.UNET_CONFIG_LARGE, x[rows], 500, enc[rows], mask[rows], prefix="unet.")
    scale = ref.abs().max().item()
    e_two = (outs[0][rows] - ref).abs().max().item() / scale
    e_f = (outs[1][rows] - ref).abs().max().item() / scale
    d = (outs[0] - outs[1]).abs().max().item() / scale
    print("UNet forward %s B2=%d: two-GEMM feed-forward vs oracle %.3e, fused vs oracle %.3e, fused vs two-GEMM %.3e" % (dtype, B2, e_two, e_f, d))
    assert e_f <= tol and e_two <= tol and d <= tol
