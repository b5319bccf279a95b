"""Mustango's Music UNet (VERDICT r2 missing #1) on the engine: UNet2DConditionModelMusic.forward
(mustango/diffusers/src/diffusers/models/unet_2d_condition_music.py:536-757) -- text, beat and chord cross-attention transformers
at every site -- and the CFG loop of MusicAudioDiffusion.inference (mustango/models.py:540-598), against the CPU oracle
(pinned to the imported class, tests/test_reference_diff.py) and against committed outputs of the imported class itself."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tango_oracle as O  # noqa: E402  (checker only)
from oracle.make_golden import music_inputs  # noqa: E402
from tango_amd import weights as W  # noqa: E402
from tango_amd.engine import Engine  # noqa: E402
from tango_amd.models import MusicAudioDiffusion  # noqa: E402
from tango_amd.scheduler import SD21_SCHEDULER_CONFIG, DDPMScheduler  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "music_unet_ref.npz")
_KEYS = ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule", "prediction_type", "clip_sample", "variance_type")


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-3), ("fp16", 3e-2), ("bf16", 1.2e-1)])
def test_music_unet_forward_tiny(dtype, tol):
    cfg = O.UNET_CONFIG_MUSIC_TINY
    e = Engine(unet=cfg, dtype=dtype)
    assert len(e.weight_names()) == len(W.unet_param_shapes(cfg, "unet."))
    # the same tensors as the reference fixture was made with: oracle/make_golden.py draws them under the UNPREFIXED key names
    sd = W.synth_state_dict(W.unet_param_shapes(cfg), 1234)
    e.load_state_dict({"unet." + k: v for k, v in sd.items()})
    e.finalize()
    x, enc, beat, chord, em, bm, cm = music_inputs(cfg, 4, 3)
    out = e.unet_forward(x.cuda(), 801, enc.cuda(), em.cuda(), beat.cuda(), chord.cuda(), bm.cuda(), cm.cuda()).cpu()
    with torch.no_grad():
        ref = O.unet_forward(sd, cfg, x, 801, enc, em, "", beat_features=beat, chord_features=chord, beat_attention_mask=bm,
                             chord_attention_mask=cm)
    err = ((out - ref).abs().max() / ref.abs().max()).item()
    print("Music UNet (tiny) %s: rel err vs oracle %.3e" % (dtype, err))
    assert err <= tol
    if dtype == "fp32":      # the imported UNet2DConditionModelMusic's own output (oracle/make_golden.py music), no oracle in the loop
        gold = np.load(GOLD)
        assert int(gold["n_tensors"]) == len(sd)
        d = np.abs(out[:, :, ::9, ::3].numpy() - gold["out_slice"]).max() / np.abs(gold["out_slice"]).max()
        cs = np.asarray([float(out.double().sum()), float(out.double().abs().sum()), float((out.double() ** 2).sum())])
        print("Music UNet (tiny) fp32 vs reference fixture: slice rel err %.3e" % d)
        assert d <= 1e-3 and np.allclose(cs[1:], gold["out_checksum"][1:], rtol=1e-3)
        # conditions matter and are routed to the right transformers: swapping beats and chords (equal lengths here) changes the output
        sw = e.unet_forward(x.cuda(), 801, enc.cuda(), em.cuda(), chord[:, :20].cuda(), beat[:, :20].cuda(), cm.cuda(), bm[:, :20].cuda()).cpu()
        base = e.unet_forward(x.cuda(), 801, enc.cuda(), em.cuda(), beat[:, :20].cuda(), chord.cuda(), bm[:, :20].cuda(), cm.cuda()).cpu()
        assert (sw - base).abs().max().item() > 1e-3
    with pytest.raises(ValueError):
        e.unet_forward(x.cuda(), 801, enc.cuda(), em.cuda())                      # a Music UNet needs its two extra conditions
    plain = Engine(unet=O.UNET_CONFIG_TINY, dtype=dtype)
    plain.load_synthetic(1234)
    with pytest.raises(ValueError):
        plain.unet_forward(x.cuda(), 801, enc.cuda(), em.cuda(), beat.cuda(), chord.cuda())


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-2), ("fp16", 1e-1)])
def test_music_denoise_loop_tiny(dtype, tol):
    """mustango/models.py:563-598: CFG loop, three conditions ordered [uncond; cond], injected noise, hipGraph == eager bitwise"""
    cfg = O.UNET_CONFIG_MUSIC_TINY
    m = MusicAudioDiffusion(unet_config=cfg, dtype=dtype)
    sd = W.synth_state_dict(W.unet_param_shapes(cfg), 1234)
    m.load_state_dict({"unet." + k: v for k, v in sd.items()})
    B, N = 2, 3
    _, enc, beat, chord, em, bm, cm = music_inputs(cfg, 2 * B, 11)
    g = torch.Generator().manual_seed(12)
    lat0 = torch.randn(B, 8, 256, 16, generator=g)
    noises = torch.randn(N, B, 8, 256, 16, generator=g)
    sch = DDPMScheduler.from_config({k: SD21_SCHEDULER_CONFIG[k] for k in _KEYS})
    outs = []
    for graph in (True, False):
        m.use_graph = graph
        outs.append(m.inference_from_embeddings(enc, em, sch, N, 3.0, latents=lat0, noise=noises.cuda(), encoded_beats=beat,
                                                beat_mask=bm, encoded_chords=chord, chord_mask=cm).cpu())
    assert torch.equal(outs[0], outs[1]), "captured graph and eager launches must agree bit for bit"
    with torch.no_grad():
        ref = O.denoise_loop(sd, cfg, O.DDPMOracle(**O.SD21_SCHEDULER), enc, em, lat0.clone(), N, 3.0, noises=list(noises),
                             music=dict(beat_features=beat, chord_features=chord, beat_attention_mask=bm, chord_attention_mask=cm))
    err = (outs[0] - ref).abs().max().item()
    print("Music denoise loop (tiny, %s, 3 CFG steps): latents max abs err %.3e" % (dtype, err))
    assert err <= tol
    with pytest.raises(ValueError):
        m.inference_from_embeddings(enc, em, sch, N, 3.0, latents=lat0)
    with pytest.raises(ValueError):
        MusicAudioDiffusion(unet_config=O.UNET_CONFIG_TINY, dtype=dtype)


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-3), ("fp16", 3e-2)])
def test_music_unet_forward_full_size(dtype, tol):
    """mustango/configs/music_diffusion_model_config.json at its real widths (1518 tensors, 1.4 G parameters), beat_len 50 /
    chord_len 20 (mustango/models.py:336,340), one CFG pair"""
    cfg = O.UNET_CONFIG_MUSIC
    e = Engine(unet=cfg, dtype=dtype)
    sd = W.synth_state_dict(W.unet_param_shapes(cfg), 1234)
    assert len(sd) == 1518
    for k in e.weight_names():                         # tensor by tensor: 1.4 G parameters
        e.set_weight(k, sd[k[len("unet."):]])
    e.finalize()
    x, enc, beat, chord, em, bm, cm = music_inputs(cfg, 2, 5, L=64)
    out = e.unet_forward(x.cuda(), 500, enc.cuda(), em.cuda(), beat.cuda(), chord.cuda(), bm.cuda(), cm.cuda()).cpu()
    with torch.no_grad():
        ref = O.unet_forward(sd, cfg, x, 500, enc, em, "", beat_features=beat, chord_features=chord, beat_attention_mask=bm,
                             chord_attention_mask=cm)
    err = ((out - ref).abs().max() / ref.abs().max()).item()
    print("Music UNet (full size, %s): rel err vs oracle %.3e" % (dtype, err))
    assert err <= tol
