"""Single-key cross-attention shortcut (round 4 prototype): samples whose text mask keeps exactly ONE key
(the unconditional half of a CFG batch: T5("") padded, models.py:282-289) get attn2 + residual as `x + to_out(v_key) + b`, the rest of
the batch runs the normal cross-attention.  The premise is pinned on the oracle in tests/test_oracle_golden.py; here the engine with the
shortcut must still match the oracle for every mask structure that selects (or must NOT select) it."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tango_oracle as O  # noqa: E402  (checker only)
from tango_amd import weights as W  # noqa: E402
from tango_amd.engine import Engine  # noqa: E402

pytestmark = pytest.mark.gpu


def _masks(kind, B2, L):
    m = torch.ones(B2, L, dtype=torch.bool)
    if kind == "cfg":                    # first half: one key each, at different positions
        for b in range(B2 // 2):
            m[b] = False
            m[b, (3 * b) % L] = True
        m[B2 // 2, L - 3:] = False       # a ragged conditional row
    elif kind == "all":                  # every sample single-key
        m[:] = False
        for b in range(B2):
            m[b, (b + 1) % L] = True
    elif kind == "short_prefix":         # only ONE leading single-key sample of four: no plan shape -> normal path
        m[0] = False
        m[0, 0] = True
    elif kind == "two_keys":             # two valid keys are not a single key
        m[: B2 // 2, 2:] = False
    return m


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-4), ("fp16", 3e-2)])
@pytest.mark.parametrize("kind", ["cfg", "all", "short_prefix", "two_keys"])
def test_unet_forward_with_single_key_rows(kind, dtype, tol):
    cfg = O.UNET_CONFIG_TINY
    sd = W.synth_state_dict(W.unet_param_shapes(cfg, "unet."), 1234)
    e = Engine(unet=cfg, dtype=dtype)
    e.load_synthetic(1234)
    g = torch.Generator().manual_seed(31)
    B2, L = 4, 16
    x = torch.randn(B2, 8, 256, 16, generator=g)
    enc = torch.randn(B2, L, cfg["cross_attention_dim"], generator=g)
    mask = _masks(kind, B2, L)
    ref = O.unet_forward(sd, cfg, x, 601, enc, mask, prefix="unet.")
    out = e.unet_forward(x.cuda(), 601, enc.cuda(), mask.cuda()).cpu()
    err = ((out - ref).abs().max() / ref.abs().max()).item()
    print("single-key structure %-12s %s: rel err vs oracle %.3e" % (kind, dtype, err))
    assert err <= tol


def test_full_size_level0_uses_the_fused_block_on_the_conditional_half():
    """866M UNet, fp16: the conditional half goes through xattn_block_kernel with offset K / V / bias pointers, the unconditional half
    through the broadcast add; both against the fp32 oracle"""
    cfg = O.UNET_CONFIG_LARGE
    sd = W.synth_state_dict(W.unet_param_shapes(cfg, "unet."), 1234)
    e = Engine(unet=cfg, dtype="fp16")
    e.load_synthetic(1234)
    g = torch.Generator().manual_seed(32)
    x = torch.randn(2, 8, 256, 16, generator=g)
    enc = torch.randn(2, 64, 1024, generator=g)
    mask = torch.ones(2, 64, dtype=torch.bool)
    mask[0, 1:] = False
    ref = O.unet_forward(sd, cfg, x, 995, enc, mask, prefix="unet.")
    out = e.unet_forward(x.cuda(), 995, enc.cuda(), mask.cuda()).cpu()
    err = ((out - ref).abs().max() / ref.abs().max()).item()
    print("full-size fp16 UNet forward with the single-key shortcut: rel err vs oracle %.3e" % err)
    assert err <= 1e-2


def test_single_key_constant_in_the_to_out_epilogue_equals_the_separate_pass():
    """At B = 8 (UNet batch 16) attn1's to_out + residual runs on the 256 x 160 / 256 x 320 GEMMs, whose epilogue adds the single-key
    rows' constant and writes them in place (GemmParams::rowvec) -- the separate rowbias_add pass must disappear from the step's
    program and the output must agree with the TANGO_NO_ROWVEC_FUSE=1 engine to fp16 rounding (one rounding fewer per site)."""
    cfg = O.UNET_CONFIG_LARGE
    g = torch.Generator().manual_seed(33)
    B2, L = 16, 64
    x = torch.randn(B2, 8, 256, 16, generator=g).cuda()
    enc = torch.randn(B2, L, 1024, generator=g).cuda()
    mask = torch.ones(B2, L, dtype=torch.bool)
    mask[: B2 // 2] = False
    mask[: B2 // 2, 0] = True
    outs, labels = {}, {}
    for fuse in (1, 0):
        os.environ["TANGO_NO_ROWVEC_FUSE"] = "0" if fuse else "1"
        e = Engine(unet=cfg, dtype="fp16")
        e.lib.tango_tuning_reload()
        e.load_synthetic(1234)
        outs[fuse] = e.unet_forward(x, 700, enc, mask.cuda()).float().cpu()
        labels[fuse] = [r[0] for r in e.profile_unet(B2, L)]
        del e
    os.environ.pop("TANGO_NO_ROWVEC_FUSE")
    from tango_amd import _lib
    _lib.load().tango_tuning_reload()
    n_sep = {k: sum(1 for lab in v if lab.startswith("xattn_single_key")) for k, v in labels.items()}
    print("separate single-key passes per step: fused %d, not fused %d" % (n_sep[1], n_sep[0]))
    # fused: the mid block (1024 rows at this batch) runs on the small tiles: its pass stays; and (round 5) the FIRST transformer of a
    # guidance step runs its part in front of attn2 once for both halves (CFG-shared prefix), so its unconditional rows are written by
    # the separate pass that reads the shared tensor (profile_unet reports the plan denoise() runs)
    assert n_sep[0] == 16 and n_sep[1] <= 2
    d = ((outs[1] - outs[0]).abs().max() / outs[0].abs().max()).item()
    print("fused vs separate single-key constant: rel diff %.3e" % d)
    assert torch.isfinite(outs[1]).all() and d <= 3e-3
