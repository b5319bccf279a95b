"""Slab form of the GroupNorm (norm.hip gn_slab_kernel, round 6): UNet levels 2-3 (<= 256 rows per sample, groups of whole 16-byte vectors) -- one
workgroup keeps a rows x (20 vectors) slab of one sample in registers: one launch, one read, one write.  Reference op: torch.nn.GroupNorm (+ SiLU) of
ResnetBlock2D.norm1 / norm2 and Transformer2DModel.norm (diffusers resnet.py:549-597, transformer_2d.py:255-262).  Checked against torch's group_norm
in fp64 on the full tensor, against the other GroupNorm forms of the engine (TANGO_GN_SLAB=0), for run-to-run bit identity, and for BATCH INVARIANCE: the
kernel choice is a function of the shape only, so a sample's result must not depend on what it is batched with."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from test_gn_coop_gpu import DT, TOL, gn, q, tuning

pytestmark = pytest.mark.gpu

# (samples, channels, rows): UNet levels 3 (8 x 8), 2 (16 x 16) and their concatenations on the up path at B2 = 2 / 16 / 64; 128 rows: the 8-vector form
SHAPES = [(2, 1280, 64), (16, 1280, 64), (64, 1280, 64), (16, 2560, 64), (16, 1280, 256), (64, 1280, 256), (3, 2560, 256), (5, 1280, 128)]


@pytest.mark.parametrize("dtype", ["fp16", "bf16", "fp32"])
@pytest.mark.parametrize("B,Cc,HW", SHAPES)
def test_gn_slab_matches_torch_and_the_other_forms(lib, dtype, B, Cc, HW):
    g = torch.Generator().manual_seed(B + Cc + HW)
    # per-channel offsets and scales so that the groups' means / variances differ by far more than the tolerance
    x = q(torch.randn(B, Cc, HW, generator=g) * (0.5 + 2.0 * torch.rand(1, Cc, 1, generator=g)) + 1.5 * torch.randn(1, Cc, 1, generator=g), dtype).cuda()
    ga, be = torch.randn(Cc, generator=g).cuda(), torch.randn(Cc, generator=g).cuda()
    for eps, act in ((1e-5, 1), (1e-6, 0)):
        ref = F.group_norm(x.double(), 32, ga.double(), be.double(), eps)
        if act:
            ref = F.silu(ref)
        ref = ref.float()
        with tuning(lib, TANGO_GN_SLAB=1):
            first = gn(lib, dtype, x, ga, be, eps, act)
            for rep in range(5):
                assert torch.equal(gn(lib, dtype, x, ga, be, eps, act), first), "repetition %d differs" % rep
            # batch invariance: the first sample alone, and the last two samples as a batch of their own
            assert torch.equal(gn(lib, dtype, x[:1].contiguous(), ga, be, eps, act), first[:1])
            assert torch.equal(gn(lib, dtype, x[-2:].contiguous(), ga, be, eps, act), first[-2:])
        with tuning(lib, TANGO_GN_SLAB=0):
            other = gn(lib, dtype, x, ga, be, eps, act)
        sc = ref.abs().max().item()
        e1 = (first - ref).abs().max().item() / sc
        e0 = (other - ref).abs().max().item() / sc
        d = (first - other).abs().max().item() / sc
        assert e1 <= TOL[dtype], "gn_slab %s B=%d C=%d rows=%d: rel err %.3e (other forms %.3e)" % (dtype, B, Cc, HW, e1, e0)
        assert d <= TOL[dtype] / 2, "gn_slab vs the other forms: %.3e" % d
        if dtype == "fp32":
            assert e1 <= 2e-6


def test_gn_slab_is_what_runs_and_is_faster(lib):
    """the slab form against the forms it replaces at the benchmarked batch (B2 = 64), HIP-event timed through the op (includes the op's layout conversions on
    both arms: a lower bound of the kernel-level gain)"""
    for Cc, HW in ((1280, 64), (1280, 256), (2560, 64)):
        g = torch.Generator().manual_seed(Cc + HW)
        x = q(torch.randn(64, Cc, HW, generator=g), "fp16").cuda()
        ga, be = torch.randn(Cc, generator=g).cuda(), torch.randn(Cc, generator=g).cuda()
        ms = {}
        for on in (0, 1, 0, 1):
            with tuning(lib, TANGO_GN_SLAB=on):
                gn(lib, "fp16", x, ga, be, 1e-5, 1)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(20):
                    gn(lib, "fp16", x, ga, be, 1e-5, 1)
                e1.record()
                torch.cuda.synchronize()
                ms[on] = min(ms.get(on, 1e9), e0.elapsed_time(e1) / 20)
        print("groupnorm op C=%d rows=%d B2=64 fp16: other forms %.3f ms, slab %.3f ms per call (op-level, conversions included)" % (Cc, HW, ms[0], ms[1]))
        assert ms[1] <= ms[0] * 1.2
