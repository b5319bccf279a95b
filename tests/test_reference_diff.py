"""Live differential tests against the imported reference (build container only; skipped on the GPU box,
where /root/reference does not exist -- the committed fixtures in tests/golden/ carry the same evidence)."""
import numpy as np
import pytest
import torch

from oracle import ref_import as R
from oracle import tango_oracle as O
from tango_amd import weights as W

pytestmark = [pytest.mark.reference, pytest.mark.skipif(not R.available(), reason="/root/reference not present")]
torch.set_grad_enabled(False)


def test_param_inventories_match_reference_state_dicts():
    with torch.device("meta"):
        unet = R.unet_cls()(**R.unet_config())
    ref = {k: tuple(v.shape) for k, v in unet.state_dict().items()}
    assert ref == {k: tuple(v) for k, v in W.unet_param_shapes(O.UNET_CONFIG_LARGE).items()}
    vae = R.autoencoder_cls()(**R.vae_config())
    enc_ref = {k: tuple(v.shape) for k, v in vae.state_dict().items() if k.startswith(("encoder", "quant_conv"))}
    assert enc_ref == dict(W.vae_encoder_param_shapes(O.VAE_CONFIG))
    ref = {k: tuple(v.shape) for k, v in vae.state_dict().items() if not k.startswith(("encoder", "quant_conv"))}
    mine = dict(W.vae_decoder_param_shapes(O.VAE_CONFIG))
    mine.update(W.hifigan_param_shapes(O.HIFIGAN_CONFIG))
    assert ref == {k: tuple(v) for k, v in mine.items()}


def test_unet_tiny_matches_fork_with_ragged_mask():
    cfgo = O.UNET_CONFIG_TINY
    cfg = dict(R.unet_config())
    cfg.update({k: cfgo[k] for k in ("block_out_channels", "attention_head_dim", "cross_attention_dim")})
    sd = W.synth_state_dict(W.unet_param_shapes(cfgo), 7)
    unet = R.unet_cls()(**cfg).eval()
    unet.load_state_dict(sd)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 8, 256, 16, generator=g)
    enc = torch.randn(3, 11, cfgo["cross_attention_dim"], generator=g)
    mask = torch.ones(3, 11, dtype=torch.bool)
    mask[0, 1:] = False
    mask[2, 4:] = False
    ref = unet(x, torch.tensor(123), encoder_hidden_states=enc, encoder_attention_mask=mask).sample
    out = O.unet_forward(sd, cfgo, x, 123, enc, mask)
    assert (ref - out).abs().max().item() < 2e-5
    # perturbing masked text tokens must not change the output (exp(-10000) == 0 in fp32)
    enc2 = enc.clone()
    enc2[2, 4:] += 5.0
    assert torch.equal(O.unet_forward(sd, cfgo, x, 123, enc2, mask), out)


def test_scheduler_step_matches_fork_bitwise():
    D = R.ddpm_cls()
    cfg = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
               prediction_type="v_prediction", clip_sample=False, variance_type="fixed_small")
    a, b = D(**cfg), O.DDPMOracle(**cfg)
    a.set_timesteps(200)
    b.set_timesteps(200)
    g = torch.Generator().manual_seed(0)
    x, v = torch.randn(2, 8, 256, 16, generator=g), torch.randn(2, 8, 256, 16, generator=g)
    for t in (995, 500, 5, 0):
        ra = a.step(v, t, x, generator=torch.Generator().manual_seed(9)).prev_sample
        rb = b.step(v, t, x, generator=torch.Generator().manual_seed(9))
        assert torch.equal(ra, rb)


def _music_ref(cfgo):
    import json
    import os
    cfg = json.load(open(os.path.join(R.REF, "mustango", "configs", "music_diffusion_model_config.json")))
    cfg = {k: v for k, v in cfg.items() if not k.startswith("_")}
    cfg.update({k: cfgo[k] for k in ("block_out_channels", "attention_head_dim", "cross_attention_dim")})
    return cfg


from oracle.make_golden import music_inputs  # noqa: E402


def test_music_unet_inventory_and_forward_match_mustango():
    """Mustango's UNet2DConditionModelMusic (unet_2d_condition_music.py:536-757): parameter inventory at the real config and the
    oracle's three-transformers-per-site forward against the imported class (tiny widths)"""
    with torch.device("meta"):
        big = R.unet_music_cls()(**_music_ref(O.UNET_CONFIG_MUSIC))
    ref = {k: tuple(v.shape) for k, v in big.state_dict().items()}
    mine = {k: tuple(v) for k, v in W.unet_param_shapes(O.UNET_CONFIG_MUSIC).items()}
    assert ref == mine and len(mine) == 1518
    cfgo = O.UNET_CONFIG_MUSIC_TINY
    torch.manual_seed(0)
    unet = R.unet_music_cls()(**_music_ref(cfgo)).eval()
    sd = W.synth_state_dict(W.unet_param_shapes(cfgo), 1234)
    unet.load_state_dict(sd)
    x, enc, beat, chord, em, bm, cm = music_inputs(cfgo, 4, 3)
    want = unet(x, torch.tensor(801), encoder_hidden_states=enc, beat_features=beat, chord_features=chord, encoder_attention_mask=em,
                beat_attention_mask=bm, chord_attention_mask=cm).sample
    got = O.unet_forward(sd, cfgo, x, 801, enc, em, "", beat_features=beat, chord_features=chord, beat_attention_mask=bm,
                         chord_attention_mask=cm)
    err = ((got - want).abs().max() / want.abs().max()).item()
    assert err <= 2e-5, err
    plain = O.unet_forward({k: v for k, v in sd.items()}, dict(cfgo, down_block_types=O.UNET_CONFIG_LARGE["down_block_types"],
                                                              up_block_types=O.UNET_CONFIG_LARGE["up_block_types"]), x, 801, enc, em, "")
    assert (plain - want).abs().max().item() > 1e-2, "the beat / chord transformers must matter"
