"""The serving callers on the REAL engine (SURVEY.md 8f rank 3 / VERDICT r2 missing #3, #5, weak #4) -- no test doubles:

  * `batch_inference.main([...])`   (reference inference.py:127-176: read prompts -> generate_for_batch -> output_{j}.wav)
  * `Predictor().setup() / .predict()`  (reference predict.py:29-67)
  * `build_pretrained_models(<AudioLDM .ckpt>)` into the real `AutoencoderKL`  (reference models.py:27-52)

Everything a caller of the reference would touch is on disk in the reference's own layout: an HF-style snapshot directory
(vae_config.json, main_config.json, pytorch_model_{main,vae}.bin) whose `text_encoder_name` is a LOCAL directory holding a
real `transformers` tokenizer (tokenizer.json, built offline with the `tokenizers` library) and a `T5EncoderModel`
(config.json + safetensors), so `Tango(path)` runs the reference's own `AutoTokenizer.from_pretrained` /
`T5EncoderModel.from_pretrained` lines (models.py:98-100) with no hub access and nothing injected.  Weights are seeded
synthetic tensors under the reference's key names (no checkpoint is reachable offline)."""
import json
import os
import shutil
import wave

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tango_oracle as O  # noqa: E402  (checker only)
from tango_amd import batch_inference as BI  # noqa: E402
from tango_amd import weights as W  # noqa: E402
from tango_amd.tango import Tango  # noqa: E402

WORDS = ["a", "dog", "barks", "twice", "rain", "wind", "in", "the", "trees", "car", "passes", "by"]


def _save_text_stack(d, seed):
    """tokenizer.json (WordLevel + '</s>' appended, the T5 convention) and a tiny random-init T5EncoderModel in directory d"""
    from tokenizers import Tokenizer, models, pre_tokenizers, processors
    from transformers import PreTrainedTokenizerFast, T5Config, T5EncoderModel
    vocab = {w: i for i, w in enumerate(["<pad>", "</s>", "<unk>"] + WORDS + ["w%d" % i for i in range(128 - 3 - len(WORDS))])}
    tk = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    tk.post_processor = processors.TemplateProcessing(single="$A </s>", special_tokens=[("</s>", 1)])
    PreTrainedTokenizerFast(tokenizer_object=tk, pad_token="<pad>", eos_token="</s>", unk_token="<unk>",
                            model_max_length=512).save_pretrained(d)
    torch.manual_seed(seed)
    cfg = T5Config(vocab_size=128, d_model=O.UNET_CONFIG_TINY["cross_attention_dim"], d_kv=64, d_ff=128, num_layers=2, num_heads=2,
                   feed_forward_proj="gated-gelu", tie_word_embeddings=False)
    enc = T5EncoderModel(cfg).eval()
    enc.save_pretrained(d)
    return enc


@pytest.fixture(scope="module")
def snapshot(tmp_path_factory):
    root = tmp_path_factory.mktemp("serving")
    d = root / "tango2"
    d.mkdir()
    text_dir = root / "flan-t5-local"
    text_dir.mkdir()
    _save_text_stack(str(text_dir), 222)                       # the hub copy: a DIFFERENT init than the checkpoint's encoder
    torch.manual_seed(111)
    from transformers import T5Config, T5EncoderModel
    ck_enc = T5EncoderModel(T5Config(vocab_size=128, d_model=O.UNET_CONFIG_TINY["cross_attention_dim"], d_kv=64, d_ff=128, num_layers=2,
                                     num_heads=2, feed_forward_proj="gated-gelu", tie_word_embeddings=False)).eval()
    ucfg = dict(O.UNET_CONFIG_TINY, _class_name="UNet2DConditionModel", _diffusers_version="0.10.0.dev0", act_fn="silu", sample_size=[32, 2])
    json.dump(ucfg, open(d / "diffusion_model_config.json", "w"))
    json.dump({"text_encoder_name": str(text_dir), "scheduler_name": "stabilityai/stable-diffusion-2-1", "unet_model_name": None,
               "unet_model_config_path": str(d / "diffusion_model_config.json"), "snr_gamma": 5.0}, open(d / "main_config.json", "w"))
    json.dump({"image_key": "fbank", "subband": 1, "embed_dim": 8, "time_shuffle": 1,
               "ddconfig": {"double_z": True, "z_channels": 8, "resolution": 256, "downsample_time": False, "in_channels": 1, "out_ch": 1,
                            "ch": 128, "ch_mult": [1, 2, 4], "num_res_blocks": 2, "attn_resolutions": [], "dropout": 0.0},
               "scale_factor": 0.9227914214134216}, open(d / "vae_config.json", "w"))
    main = dict(W.synth_state_dict(W.unet_param_shapes(O.UNET_CONFIG_TINY, "unet."), 1234))
    main.update({"text_encoder." + k: v.clone() for k, v in ck_enc.state_dict().items()})
    torch.save(main, d / "pytorch_model_main.bin")
    shapes = W.vae_decoder_param_shapes(O.VAE_CONFIG)
    shapes.update(W.hifigan_param_shapes(O.HIFIGAN_CONFIG))
    torch.save(dict(W.synth_state_dict(shapes, 1234)), d / "pytorch_model_vae.bin")
    return str(root), str(d), ck_enc


def _read_wav(path):
    with wave.open(path, "rb") as w:
        meta = (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes())
        return meta, np.frombuffer(w.readframes(w.getnframes()), "<i2")


@pytest.mark.parametrize("text_encoder", ["torch", "engine"])
def test_batch_inference_main_on_the_engine(snapshot, tmp_path, monkeypatch, text_encoder):
    root, path, ck_enc = snapshot
    prompts = ["a dog barks twice", "rain", "wind in the trees"]
    tf = tmp_path / "test.json"
    tf.write_text("\n".join(json.dumps({"captions": c, "audiocap_id": i}) for i, c in enumerate(prompts)) + "\n")
    steps, guidance = 3, 3.0
    # seed right before generation, so the run can be reproduced through the Python API below (model construction itself
    # consumes torch's global generator: T5EncoderModel.from_pretrained initialises before it loads)
    orig = BI.generate_and_save

    def seeded(gen, *a, **k):
        torch.manual_seed(31)
        return orig(gen, *a, **k)
    monkeypatch.setattr(BI, "generate_and_save", seeded)
    out_root = tmp_path / "outputs"
    rec = BI.main(["--model", path, "--test_file", str(tf), "--text_key", "captions", "--num_steps", str(steps), "--guidance", str(guidance),
                   "--batch_size", "2", "--dtype", "fp32", "--out_root", str(out_root), "--text_encoder", text_encoder])
    assert rec["Test Instances"] == 3 and rec["Steps"] == steps and abs(rec["audio_seconds"] - 3 * 163872 / 16000.0) < 1e-6
    files = sorted(os.listdir(rec["output_dir"]))
    assert files == ["output_0.wav", "output_1.wav", "output_2.wav"]
    got = []
    for j in range(3):
        meta, a = _read_wav(os.path.join(rec["output_dir"], "output_%d.wav" % j))
        assert meta == (1, 2, 16000, 163872)                 # 16 kHz mono int16, 10.24 s (hifigan/utilities.py:9-39, inference.py:150)
        got.append(a)
    assert not np.array_equal(got[0], got[1]) and all(np.abs(a.astype(np.int32)).max() > 0 for a in got)
    # the same generation through the Python API on a second Tango built from the same snapshot
    t = Tango(path, device="cuda:0", dtype="fp32", text_encoder="engine" if text_encoder == "engine" else None)
    torch.manual_seed(31)
    want = t.generate_for_batch(prompts, steps=steps, guidance=guidance, samples=1, batch_size=2)
    for j in range(3):
        assert np.array_equal(got[j], want[j]), "output_%d.wav differs from generate_for_batch" % j
    # the checkpoint's text-encoder tensors (not the local 'hub' copy's) produced the embeddings
    pe, pm = t.model.encode_text(["rain"])
    b = t.model.tokenizer(["rain"], return_tensors="pt")
    with torch.no_grad():
        ref = ck_enc(input_ids=b.input_ids, attention_mask=b.attention_mask)[0]
    assert (pe.cpu().float() - ref).abs().max().item() < 2e-4


def test_predictor_on_the_engine(snapshot, tmp_path):
    from tango_amd.predict import Predictor
    root, path, _ = snapshot
    p = Predictor()
    with pytest.raises(FileNotFoundError):
        p.setup(model_cache=root, dtype="fp32")                 # tango2-full is not there: no silent download
    shutil.copytree(path, os.path.join(root, "tango2-full"))
    p.setup(model_cache=root, dtype="fp32")
    assert sorted(p.models) == ["tango2", "tango2-full"]
    torch.manual_seed(9)
    out = p.predict("a car passes by", "tango2", 3, 3.0, out=str(tmp_path / "o.wav"))
    meta, a = _read_wav(str(out))
    assert meta == (1, 2, 16000, 163872)
    torch.manual_seed(9)
    want = p.models["tango2"].generate("a car passes by", 3, 3.0)
    assert np.array_equal(a, want)
    torch.manual_seed(9)
    b = p.models["tango2-full"].generate("a car passes by", 3, 3.0)       # same weights, own engine + own text encoder
    assert np.array_equal(a, b)
    with pytest.raises(KeyError):
        p.predict("x", "tango", 1, 1.0)


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_build_pretrained_models_audioldm_ckpt_on_the_engine(tmp_path, dtype):
    """models.py:27-52: an AudioLDM `.ckpt` file -> `first_stage_model.*` + `scale_factor` -> the real engine AutoencoderKL;
    the scale factor of the CHECKPOINT (not the config default) must divide the latents (autoencoder.py:116-124)."""
    from tango_amd.autoencoder import AutoencoderKL
    from tango_amd.models import build_pretrained_models
    shapes = W.vae_decoder_param_shapes(O.VAE_CONFIG)
    shapes.update(W.hifigan_param_shapes(O.HIFIGAN_CONFIG))
    vsd = W.synth_state_dict(shapes, 4321)
    sf = 0.6180339887
    sd = {"first_stage_model." + k: v for k, v in vsd.items()}
    sd["scale_factor"] = torch.tensor(sf)
    sd["first_stage_model.encoder.conv_in.weight"] = torch.zeros(128, 1, 3, 3)       # keys of the checkpoint the decoder path ignores
    sd["first_stage_model.loss.logvar"] = torch.zeros(())
    sd["model.diffusion_model.time_embed.0.weight"] = torch.zeros(4, 4)               # AudioLDM's own LDM: not ours
    sd["cond_stage_model.model.logit_scale_a"] = torch.zeros(())
    ck = tmp_path / "audioldm-s-full.ckpt"
    torch.save({"state_dict": sd, "global_step": 1}, ck)
    vae, stft = build_pretrained_models(str(ck), dtype=dtype)
    assert isinstance(vae, AutoencoderKL) and abs(vae.scale_factor - sf) < 1e-7 and vae.device() == torch.device("cuda:0")
    assert stft is not None and hasattr(stft, "mel_spectrogram")                      # the TacotronSTFT front-end (second item of the reference's tuple)
    g = torch.Generator().manual_seed(8)
    z = torch.randn(2, 8, 256, 16, generator=g)
    mel = vae.decode_first_stage(z.cuda())
    with torch.no_grad():
        ref = O.vae_decode_first_stage(vsd, dict(O.VAE_CONFIG, scale_factor=sf), z)
    err = ((mel.cpu() - ref).abs().max() / ref.abs().max()).item()
    print("build_pretrained_models(.ckpt) %s: decode_first_stage rel err vs oracle %.3e" % (dtype, err))
    assert mel.shape == (2, 1, 1024, 64) and err <= (1e-3 if dtype == "fp32" else 3e-2)
    wav = vae.decode_to_waveform(mel)
    assert wav.dtype == np.int16 and wav.shape == (2, 163872)
