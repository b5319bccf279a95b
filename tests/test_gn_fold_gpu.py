"""GroupNorm folded into proj_in (round 6: norm.hip gn_fold_kernel + per-sample weights of the 256 x 320 / 256 x 160 GEMMs; reference ops
Transformer2DModel.norm -> proj_in, transformer_2d.py:255-262): proj_in(GroupNorm(x)) = Wf_s x + bf_s with per-sample folded weights, the
normalised tensor is never written.  Not bit-identical to the two-op path (the weights are rounded after folding instead of the
activations after normalising): both must sit within the engine's tolerance of the fp32 oracle, and close to each other."""
import pytest
import torch

from oracle import tango_oracle as O
from tango_amd import weights as W
from tango_amd.engine import UNET_CONFIG_LARGE, Engine
from test_duo_gpu import tuning

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,tol", [("fp16", 4e-3), ("bf16", 4e-2)])
def test_unet_forward_with_folded_groupnorm(lib, dtype, tol):
    B2 = 16                                              # level 0 / 1 tensors > 8 MiB: the fold applies at 10 of the 16 transformers
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B2, 8, 256, 16, generator=g)
    # a mean far from zero in some channels of the residual stream is what the fold has to cancel: shift two latent channels
    x[:, 3] += 2.5
    x[:, 6] -= 1.5
    enc = torch.randn(B2, 64, 1024, generator=g)
    mask = torch.ones(B2, 64, dtype=torch.bool)
    mask[: B2 // 2, 1:] = False
    e = Engine(unet=UNET_CONFIG_LARGE, dtype=dtype)
    e.load_synthetic(1234)
    outs = {}
    for nofold in (1, 0):
        with tuning(lib, TANGO_GN_FOLD=1 - nofold):
            e.drop_plans()
            outs[nofold] = e.unet_forward(x.cuda(), 500, enc.cuda(), mask.cuda()).cpu()
            labels = [r[0] for r in e.profile_unet(B2, 64)]
        assert any(l.startswith("groupnorm(fold)") for l in labels) == (nofold == 0), labels[:12]
    e.drop_plans()
    sd = W.synth_state_dict(W.unet_param_shapes(O.UNET_CONFIG_LARGE, "unet."), 1234)
    rows = [0, 11]
    with torch.no_grad():
        ref = O.unet_forward(sd, O.UNET_CONFIG_LARGE, x[rows], 500, enc[rows], mask[rows], prefix="unet.")
    scale = ref.abs().max().item()
    e_two = (outs[1][rows] - ref).abs().max().item() / scale
    e_fold = (outs[0][rows] - ref).abs().max().item() / scale
    d = (outs[0] - outs[1]).abs().max().item() / scale
    print("UNet forward %s B2=%d: two-op GroupNorm + proj_in vs oracle %.3e, folded vs oracle %.3e, folded vs two-op %.3e" % (dtype, B2, e_two, e_fold, d))
    assert e_fold <= tol and e_two <= tol and d <= tol
    assert not torch.equal(outs[0], outs[1])
