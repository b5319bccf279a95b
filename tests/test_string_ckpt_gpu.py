"""The string and checkpoint paths of the drop-in API, executed end to end (rows a1, a2, f2):

  snapshot dir -> vae_config.json / main_config.json -> torch.load(pytorch_model_{main,vae}.bin) -> load_state_dict
  -> Tango(path).generate("...") / generate_for_batch([...], samples=2)                    (tango.py:10-64)
  -> AudioDiffusion.inference -> encode_text_classifier_free -> tokenizer + T5EncoderModel  (models.py:210-305)

No hub access exists offline, so the snapshot is synthetic: seeded tensors under the reference's own key names, a
random-init `transformers.T5EncoderModel` (the real class, tiny config) and a whitespace tokenizer double with the
HF call signature.  The latents of the string path are compared with the CPU oracle fed the SAME embeddings, initial
latents and step noise (recovered through the seed path torch.manual_seed -> Philox key)."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tango_oracle as O  # noqa: E402  (checker only)
from tango_amd import weights as W  # noqa: E402
from tango_amd.tango import Tango  # noqa: E402


class WhitespaceTokenizer:
    """Tokenizer double with the transformers call surface used by models.py:129-147,266-305."""
    model_max_length = 512
    pad_token_id, eos_token_id = 0, 1

    def __init__(self, vocab_size=128):
        self.vocab_size = vocab_size

    def _ids(self, text):
        ids = [2 + (sum(ord(ch) * (i + 1) for i, ch in enumerate(w)) % (self.vocab_size - 2)) for w in text.split()]
        return ids + [self.eos_token_id]

    def __call__(self, texts, max_length=None, padding=True, truncation=True, return_tensors="pt"):
        rows = [self._ids(t) for t in texts]
        if truncation and max_length:
            rows = [r[:max_length] for r in rows]
        width = max_length if padding == "max_length" else max(len(r) for r in rows)
        ids = torch.full((len(rows), width), self.pad_token_id, dtype=torch.long)
        am = torch.zeros((len(rows), width), dtype=torch.long)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = torch.tensor(r)
            am[i, :len(r)] = 1
        return type("BatchEncoding", (), {"input_ids": ids, "attention_mask": am})()


def tiny_t5(seed):
    from transformers import T5Config, T5EncoderModel
    torch.manual_seed(seed)
    cfg = T5Config(vocab_size=128, d_model=O.UNET_CONFIG_TINY["cross_attention_dim"], d_kv=64, d_ff=128, num_layers=2, num_heads=2,
                   feed_forward_proj="gated-gelu", tie_word_embeddings=False)
    return T5EncoderModel(cfg).eval()


@pytest.fixture(scope="module")
def snapshot(tmp_path_factory):
    """A directory laid out like the HF snapshot `declare-lab/tango` (tango.py:12-28)."""
    d = tmp_path_factory.mktemp("tango_snapshot")
    ucfg = dict(O.UNET_CONFIG_TINY, _class_name="UNet2DConditionModel", _diffusers_version="0.10.0.dev0", act_fn="silu",
                sample_size=[32, 2])
    json.dump(ucfg, open(d / "diffusion_model_config.json", "w"))
    json.dump({"text_encoder_name": "google/flan-t5-large", "scheduler_name": "stabilityai/stable-diffusion-2-1", "unet_model_name": None,
               "unet_model_config_path": str(d / "diffusion_model_config.json"), "snr_gamma": 5.0}, open(d / "main_config.json", "w"))
    json.dump({"image_key": "fbank", "subband": 1, "embed_dim": 8, "time_shuffle": 1,
               "ddconfig": {"double_z": True, "z_channels": 8, "resolution": 256, "downsample_time": False, "in_channels": 1, "out_ch": 1,
                            "ch": 128, "ch_mult": [1, 2, 4], "num_res_blocks": 2, "attn_resolutions": [], "dropout": 0.0},
               "scale_factor": 0.9227914214134216}, open(d / "vae_config.json", "w"))
    t5_ckpt = tiny_t5(111)                                         # the encoder weights stored IN the checkpoint
    main = dict(W.synth_state_dict(W.unet_param_shapes(O.UNET_CONFIG_TINY, "unet."), 1234))
    main.update({"text_encoder." + k: v.clone() for k, v in t5_ckpt.state_dict().items()})
    torch.save(main, d / "pytorch_model_main.bin")
    shapes = W.vae_decoder_param_shapes(O.VAE_CONFIG)
    shapes.update(W.hifigan_param_shapes(O.HIFIGAN_CONFIG))
    vae = dict(W.synth_state_dict(shapes, 1234))
    vae["encoder.conv_in.weight"] = torch.zeros(128, 1, 3, 3)       # training-side keys the engine must ignore
    vae["quant_conv.weight"] = torch.zeros(16, 16, 1, 1)
    torch.save(vae, d / "pytorch_model_vae.bin")
    return str(d), t5_ckpt, main, vae


def test_tango_from_snapshot_generate_strings(snapshot, lib):
    path, t5_ckpt, main_sd, vae_sd = snapshot
    # the injected encoder starts from a DIFFERENT init: the checkpoint's text_encoder.* tensors must replace it
    t = Tango(path, device="cuda:0", dtype="fp32", text_encoder=tiny_t5(222).cuda(), tokenizer=WhitespaceTokenizer())
    assert t.scheduler.config.prediction_type == "v_prediction"
    prompts = ["a dog barks twice", "rain"]
    steps, guidance = 3, 3
    # ---- a2: encode_text_classifier_free == the checkpoint's encoder on the same token ids, [uncond; cond] order ----
    pe, pm = t.model.encode_text_classifier_free(prompts, 1)
    tok = WhitespaceTokenizer()
    b = tok(prompts, max_length=512)
    with torch.no_grad():
        cond = t5_ckpt(input_ids=b.input_ids, attention_mask=b.attention_mask)[0]
        u = tok([""] * 2, max_length=cond.shape[1], padding="max_length")
        unc = t5_ckpt(input_ids=u.input_ids, attention_mask=u.attention_mask)[0]
    assert pe.shape == (4, cond.shape[1], cond.shape[2]) and pm.dtype == torch.bool
    assert (pe.cpu() - torch.cat([unc, cond])).abs().max().item() < 1e-4, "checkpoint text-encoder weights did not take effect"
    assert pm.cpu().tolist() == torch.cat([u.attention_mask, b.attention_mask]).bool().tolist()
    assert pm[0].sum().item() == 1                                   # T5("") = [EOS, pad, ...] attends to one token

    # ---- a1: the string path end to end vs the oracle on the same embeddings / latents / step noise ----
    torch.manual_seed(77)
    latents = t.model.inference(prompts, t.scheduler, steps, guidance, 1, disable_progress=True)
    torch.manual_seed(77)
    lat0 = torch.randn(2, 8, 256, 16, device="cuda") * t.scheduler.init_noise_sigma      # models.py:259-264, same generator
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())                                  # the Philox key the shim derives
    noises = []
    for i in range(steps):
        n = torch.empty(2, 8, 4096, device="cuda")
        assert lib.tango_op_philox_normal(C.c_void_p(n.data_ptr()), 2, 8, 4096, i, seed, 0, None) == 0
        noises.append(n.view(2, 8, 256, 16).cpu())
    usd = {k: v for k, v in main_sd.items() if k.startswith("unet.")}
    with torch.no_grad():
        ref = O.denoise_loop(usd, O.UNET_CONFIG_TINY, O.DDPMOracle(**O.SD21_SCHEDULER), pe.cpu().float(), pm.cpu(), lat0.cpu(), steps,
                             guidance, noises=noises, prefix="unet.")
    err = (latents.cpu() - ref).abs().max().item()
    print("Tango(path) string path, 3 steps: latents max abs err vs oracle %.3e" % err)
    assert latents.shape == (2, 8, 256, 16) and err <= 1e-2

    # ---- generate / generate_for_batch shapes and grouping (tango.py:43-64) ----
    torch.manual_seed(5)
    wave = t.generate("a dog barks twice", steps=steps, guidance=guidance)
    assert isinstance(wave, np.ndarray) and wave.dtype == np.int16 and wave.shape == (163872,)
    torch.manual_seed(5)
    wave2 = t.generate("a dog barks twice", steps=steps, guidance=guidance)
    assert np.array_equal(wave, wave2), "torch.manual_seed fixes initial latents AND step noise"
    wave3 = t.generate("a dog barks twice", steps=steps, guidance=guidance)
    assert not np.array_equal(wave, wave3), "consecutive calls draw fresh noise"
    outs = t.generate_for_batch(["rain", "a dog barks twice", "wind in trees"], steps=2, guidance=guidance, samples=2, batch_size=2)
    assert len(outs) == 3 and all(len(g) == 2 and g[0].shape == (163872,) for g in outs)
    flat = t.generate_for_batch(["rain", "wind"], steps=2, guidance=1.0, samples=1, batch_size=8)   # no-CFG branch (models.py:213)
    assert len(flat) == 2 and flat[0].dtype == np.int16
    # waveform of the decoded oracle latents == the engine's decode of the same latents (VAE + vocoder from the .bin)
    mel = t.vae.decode_first_stage(ref.cuda())
    vv = {k: v for k, v in vae_sd.items() if not k.startswith(("encoder.", "quant_conv."))}
    with torch.no_grad():
        mel_ref = O.vae_decode_first_stage(vv, O.VAE_CONFIG, ref)
    assert ((mel.cpu() - mel_ref).abs().max() / mel_ref.abs().max()).item() <= 1e-3


def test_snapshot_errors(snapshot, tmp_path):
    """strict loading like nn.Module.load_state_dict: a missing UNet tensor or a foreign text-encoder key is an error"""
    path, _, main_sd, _ = snapshot
    import shutil
    d = tmp_path / "broken"
    shutil.copytree(path, d)
    cfgp = json.load(open(d / "main_config.json"))
    cfgp["unet_model_config_path"] = str(d / "diffusion_model_config.json")
    json.dump(cfgp, open(d / "main_config.json", "w"))
    bad = dict(main_sd)
    bad.pop("unet.conv_in.weight")
    torch.save(bad, d / "pytorch_model_main.bin")
    with pytest.raises(RuntimeError, match="Missing key"):
        Tango(str(d), dtype="fp32", text_encoder=tiny_t5(1).cuda(), tokenizer=WhitespaceTokenizer())
    bad = dict(main_sd)
    bad["text_encoder.encoder.block.7.layer.0.SelfAttention.q.weight"] = torch.zeros(64, 96)
    torch.save(bad, d / "pytorch_model_main.bin")
    with pytest.raises(RuntimeError, match="text_encoder"):
        Tango(str(d), dtype="fp32", text_encoder=tiny_t5(1).cuda(), tokenizer=WhitespaceTokenizer())


def test_tango_text_encoder_on_engine(snapshot):
    """text_encoder="engine": the checkpoint's FLAN-T5 tensors run on the HIP engine too -- no torch module, no hub access;
    embeddings equal the checkpoint encoder's (transformers, CPU fp32) and generation works end to end."""
    path, t5_ckpt, _, _ = snapshot
    t = Tango(path, device="cuda:0", dtype="fp32", text_encoder="engine", tokenizer=WhitespaceTokenizer())
    from tango_amd.text_encoder import T5EncoderOnEngine
    assert isinstance(t.model.text_encoder, T5EncoderOnEngine)
    prompts = ["a dog barks twice", "rain"]
    pe, pm = t.model.encode_text_classifier_free(prompts, 2)
    tok = WhitespaceTokenizer()
    b = tok(prompts, max_length=512)
    with torch.no_grad():
        cond = t5_ckpt(input_ids=b.input_ids, attention_mask=b.attention_mask)[0].repeat_interleave(2, 0)
        u = tok([""] * 2, max_length=cond.shape[1], padding="max_length")
        unc = t5_ckpt(input_ids=u.input_ids, attention_mask=u.attention_mask)[0].repeat_interleave(2, 0)
    want = torch.cat([unc, cond])
    valid = pm.cpu()
    err = ((pe.cpu() - want).abs()[valid].max() / want.abs()[valid].max()).item()
    print("engine T5 inside Tango: embeddings rel err vs the checkpoint encoder %.3e" % err)
    assert pe.shape == want.shape and err <= 2e-4
    torch.manual_seed(3)
    w1 = t.generate("rain", steps=2, guidance=3)
    assert w1.dtype == np.int16 and w1.shape == (163872,)
