"""512-pixel x 160-channel form of the wide 3x3 conv tile (round 6: conv3x3_wide_kernel<TALL>): the same eight 64 x 160 wave tiles,
the same MFMA order per accumulator and the same epilogue as the 256 x 320 form, so the two must agree to the last bit -- and
repeat bit-identically (halo of 512 pixels, two-deep weight DMA groups: counted waits as in the wide form).

The dispatch switches are launch-time (tuning.h): the tests set the environment, call tango_tuning_reload(), and restore it."""
import os

import pytest
import torch
import torch.nn.functional as F

from test_duo_gpu import DT, TOL, p, q, run, tuning

pytestmark = pytest.mark.gpu

REPS = int(os.environ.get("TANGO_TALL_REPS", "12"))
TALL = dict(TANGO_CONV_TALL=1, TANGO_CONV_PIPE=0, TANGO_FORCE_DMA_GEMM=1)
WIDE = dict(TANGO_CONV_TALL=0, TANGO_CONV_PIPE=0, TANGO_FORCE_DMA_GEMM=1)


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("B,Cin,H,W,Cout,ups", [
    (2, 128, 256, 16, 320, 0),        # 16 tall tiles, four channel chunks
    (4, 320, 256, 16, 320, 0),        # level-0 ResBlock conv
    (4, 960, 256, 16, 320, 0),        # level-0 up path (K = 8640)
    (8, 640, 128, 8, 640, 0),         # level 1: four column tiles, halo of 64 image rows
    (16, 1280, 64, 4, 1280, 0),       # level 2: two images per tile
    (16, 1920, 64, 4, 640, 0),        # N = 640 at level-2 geometry
    (4, 640, 128, 8, 640, 1),         # fused nearest x2 upsampling onto the level-0 grid
    (8, 1280, 64, 4, 1280, 1),        # ... onto the level-1 grid
    (3, 320, 256, 16, 320, 0),        # M = 12288 = 24 tall tiles (odd number of 256-pixel tiles per image pair is impossible; odd batch)
])
def test_conv_tall_bit_equal(lib, dtype, B, Cin, H, W, Cout, ups):
    g = torch.Generator().manual_seed(B + Cin + H + Cout + ups)
    x = q(torch.randn(B, Cin, H, W, generator=g), dtype).cuda()
    w = q(torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5, dtype).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    Ho, Wo = H << ups, W << ups
    call = lambda out: lib.tango_op_conv2d(DT[dtype], p(x), p(w), p(b), p(out), B, Cin, H, W, Cout, 1, ups, None)
    with tuning(lib, **WIDE):
        ref = run(lib, call, (B, Cout, Ho, Wo))
    with tuning(lib, **TALL):
        out = run(lib, call, (B, Cout, Ho, Wo), REPS)
    assert torch.equal(out, ref), "tall vs wide conv tile: %d elements differ" % (out != ref).sum().item()
    h = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x, w, b, padding=1)
    err = ((out - h).abs().max() / (h.abs().max() + 1e-9)).item()
    assert err <= TOL[dtype], err


def test_unet_forward_tall_equals_wide(lib):
    """the whole UNet (residual / time-bias epilogues, concat column slices as conv inputs and outputs) at B2 = 16, the production
    routing (level 0 on the wide conv: 256 tiles): bit-equal"""
    from tango_amd.engine import UNET_CONFIG_LARGE, Engine
    g = torch.Generator().manual_seed(11)
    x = torch.randn(16, 8, 256, 16, generator=g).cuda()
    enc = torch.randn(16, 64, 1024, generator=g).cuda()
    mask = torch.ones(16, 64, dtype=torch.bool).cuda()
    e = Engine(unet=UNET_CONFIG_LARGE, dtype="fp16")
    e.load_synthetic(1234)
    outs = []
    for env in (dict(TANGO_CONV_TALL=0, TANGO_CONV_PIPE=0), dict(TANGO_CONV_TALL=1), dict(TANGO_CONV_TALL=0, TANGO_CONV_PIPE=2)):
        with tuning(lib, **env):
            e.drop_plans()
            outs.append(e.unet_forward(x, 500, enc, mask).clone())
    e.drop_plans()
    assert torch.equal(outs[0], outs[1]), "%d elements differ" % (outs[0] != outs[1]).sum().item()
    assert torch.equal(outs[0], outs[2]), "default (in-step pipelined) conv vs ping-pong: %d elements differ" % (outs[0] != outs[2]).sum().item()
