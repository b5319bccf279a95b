"""`AudioDiffusion` -- drop-in for the inference-side surface of the reference class
(models.py:55-305 of declare-lab/tango): `inference`, `prepare_latents`, `encode_text`,
`encode_text_classifier_free`, `.unet.config.in_channels`.  Training (`forward`, `compute_snr`) is
out of scope.

The denoise loop (models.py:233-249) runs inside the HIP engine; the frozen FLAN-T5 encoder stays a
PyTorch-ROCm module (its output is the engine's input).
"""
import json
import os
from types import SimpleNamespace
from typing import List, Optional

import torch

from .engine import Engine, normalize_unet_config
from .scheduler import DDIMScheduler, DDPMScheduler  # noqa: F401  (re-exported like `from models import DDPMScheduler`)

_TEXT_BUCKETS = (16, 32, 64, 128, 256, 512)


def build_pretrained_models(ckpt, vae_config=None, dtype="fp16", device="cuda:0", autoencoder_cls=None, stft_config=None, stft_cls=None):
    """models.py:27-52 of the reference: build the mel-VAE (+ vocoder) and the TacotronSTFT wave->mel front-end from an AudioLDM
    `.ckpt` (`{"state_dict": {"first_stage_model.*": ..., "scale_factor": ...}}`); `ckpt` is a path or an already loaded dict.
    Returns `(vae, fn_STFT)` like the reference.  `vae_config` defaults to the AudioLDM first-stage config
    (audioldm/utils.py:158-181 == mustango/configs/vae_config.json), `stft_config` to its preprocessing block
    (audioldm/utils.py:104-118: n_fft 1024, hop 160, 64 mels, 16 kHz, 0-8000 Hz)."""
    if autoencoder_cls is None:
        from .autoencoder import AutoencoderKL as autoencoder_cls
    if stft_cls is None:
        from .stft import TacotronSTFT as stft_cls
    checkpoint = torch.load(ckpt, map_location="cpu") if isinstance(ckpt, (str, os.PathLike)) else ckpt
    sd = checkpoint["state_dict"]
    scale_factor = float(sd["scale_factor"].item() if hasattr(sd["scale_factor"], "item") else sd["scale_factor"])
    vae_state_dict = {k[len("first_stage_model."):]: v for k, v in sd.items() if k.startswith("first_stage_model.")}
    cfg = dict(vae_config) if vae_config is not None else dict(
        image_key="fbank", subband=1, embed_dim=8, time_shuffle=1,
        ddconfig=dict(double_z=True, z_channels=8, resolution=256, downsample_time=False, in_channels=1, out_ch=1, ch=128,
                      ch_mult=[1, 2, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0))
    cfg["scale_factor"] = scale_factor
    vae = autoencoder_cls(**cfg, dtype=dtype, device=device)
    vae.load_state_dict(vae_state_dict)
    sc = dict(stft_config) if stft_config is not None else dict(filter_length=1024, hop_length=160, win_length=1024, n_mel_channels=64,
                                                                 sampling_rate=16000, mel_fmin=0, mel_fmax=8000)
    fn_STFT = stft_cls(sc["filter_length"], sc["hop_length"], sc["win_length"], sc["n_mel_channels"], sc["sampling_rate"],
                       sc["mel_fmin"], sc["mel_fmax"], device=device)
    vae.eval()
    fn_STFT.eval()
    return vae, fn_STFT


def _hf_classes():
    """(AutoTokenizer, T5EncoderModel) of models.py:98-100, imported on first use"""
    from transformers import AutoTokenizer, T5EncoderModel
    return AutoTokenizer, T5EncoderModel


class _UNetHandle:
    """What callers read off `model.unet` (models.py:227)."""

    def __init__(self, cfg, engine):
        self.config = SimpleNamespace(**cfg)
        self.in_channels = cfg["in_channels"]
        self._engine = engine

    def __call__(self, sample, timestep, encoder_hidden_states=None, encoder_attention_mask=None, beat_features=None,
                 chord_features=None, beat_attention_mask=None, chord_attention_mask=None, **_):
        out = self._engine.unet_forward(sample, int(timestep), encoder_hidden_states, encoder_attention_mask, beat_features,
                                        chord_features, beat_attention_mask, chord_attention_mask)
        return SimpleNamespace(sample=out)


class AudioDiffusion:
    def __init__(self, text_encoder_name=None, scheduler_name=None, unet_model_name=None, unet_model_config_path=None,
                 snr_gamma=None, freeze_text_encoder=True, uncondition=False, *, unet_config: Optional[dict] = None,
                 dtype: str = "fp16", device="cuda:0", text_encoder=None, tokenizer=None, bucket_text_len: bool = True,
                 attn_fp8: bool = False):
        assert unet_model_name is None, "released Tango checkpoints take the set_from == 'random' branch (models.py:83-86)"
        if unet_config is None:
            if unet_model_config_path is None:
                raise ValueError("Either UNet pretrain model name or a config file path is required")
            unet_config = json.load(open(unet_model_config_path))
        self.unet_config = normalize_unet_config({k: v for k, v in unet_config.items() if not k.startswith("_")})
        self.text_encoder_name = text_encoder_name
        self.scheduler_name = scheduler_name
        self.set_from = "random"
        self.device = torch.device(device)
        self.engine = Engine(unet=self.unet_config, dtype=dtype, device=device, attn_fp8=attn_fp8)
        self.unet = _UNetHandle(self.unet_config, self.engine)
        self.text_encoder = text_encoder
        self.tokenizer = tokenizer
        self.bucket_text_len = bucket_text_len
        self.use_graph = True
        self.seed = None          # None: every call draws its Philox key from torch's default generator (torch.manual_seed rules)
        self._calls = 0
        self._text_sd = None      # `text_encoder.*` tensors of pytorch_model_main.bin, applied when the encoder exists

    # ---- state dict --------------------------------------------------------------------------
    def load_state_dict(self, sd, strict=True):
        """pytorch_model_main.bin: `unet.*` goes to the engine, `text_encoder.*` to the torch T5."""
        missing = self.engine.load_state_dict(sd, strict=strict)
        self.engine.finalize()
        # The reference loads the checkpoint's text encoder through the same call (tango.py:28 -> nn.Module.load_state_dict,
        # strict): checkpoints trained with freeze_text_encoder=False carry tuned T5 weights.  The encoder may not exist yet
        # (it is built lazily), so the sub-dict is kept and applied by _ensure_text().
        te = {k[len("text_encoder."):]: v for k, v in sd.items() if k.startswith("text_encoder.")}
        self._text_sd = te or None
        if isinstance(self.text_encoder, str):
            # text_encoder="engine": run FLAN-T5 on the HIP engine too (tango_amd/text_encoder.py), built from the checkpoint's
            # own tensors -- no hub access, no PyTorch compute left in generate()
            if self.text_encoder != "engine":
                raise ValueError("text_encoder must be a module, None or the string 'engine'")
            if not te:
                raise RuntimeError("text_encoder='engine' needs the checkpoint's text_encoder.* tensors")
            from .text_encoder import T5EncoderOnEngine
            self.text_encoder = T5EncoderOnEngine.from_state_dict(sd, "text_encoder.", device=self.device)
            self._text_sd = None
        elif self.text_encoder is not None:
            self._apply_text_sd()
        return missing

    def _apply_text_sd(self):
        if self._text_sd is None:
            return
        res = self.text_encoder.load_state_dict(self._text_sd, strict=False)
        # `shared.weight` and `encoder.embed_tokens.weight` are one tied tensor: either name alone is complete
        tied = {"shared.weight", "encoder.embed_tokens.weight"}
        miss = [k for k in res.missing_keys if not (k in tied and tied & set(self._text_sd))]
        if miss or res.unexpected_keys:
            raise RuntimeError("text_encoder state_dict mismatch: missing %s unexpected %s" % (miss[:4], list(res.unexpected_keys)[:4]))
        self._text_sd = None

    def eval(self):
        return self

    def to(self, device):
        return self

    # ---- text --------------------------------------------------------------------------------
    def _ensure_text(self):
        if isinstance(self.text_encoder, str):
            raise RuntimeError("text_encoder='engine' is built by load_state_dict(); no checkpoint has been loaded yet")
        # each component is loaded only if IT is missing (models.py:98-100): an encoder that load_state_dict() already built
        # from the checkpoint (text_encoder="engine") or that the caller supplied must never be replaced by stock hub
        # weights just because no tokenizer was passed
        if self.tokenizer is None:
            self.tokenizer = _hf_classes()[0].from_pretrained(self.text_encoder_name)
        if self.text_encoder is None:
            self.text_encoder = _hf_classes()[1].from_pretrained(self.text_encoder_name).to(self.device).eval()
        self._apply_text_sd()

    def encode_text(self, prompt: List[str]):
        """models.py:129-147"""
        self._ensure_text()
        batch = self.tokenizer(prompt, max_length=self.tokenizer.model_max_length, padding=True, truncation=True,
                               return_tensors="pt")
        ids, am = batch.input_ids.to(self.device), batch.attention_mask.to(self.device)
        with torch.no_grad():
            hs = self.text_encoder(input_ids=ids, attention_mask=am)[0]
        return hs, (am == 1).to(self.device)

    def encode_text_classifier_free(self, prompt: List[str], num_samples_per_prompt: int):
        """models.py:266-305: returns [uncond; cond] embeddings and the boolean mask."""
        pe, pm, _ = self._encode_text_classifier_free(prompt, num_samples_per_prompt)
        return pe, pm

    def _encode_text_classifier_free(self, prompt: List[str], num_samples_per_prompt: int):
        """the same, plus the tokenizer's own HOST copy of the mask (third value): `inference` hands it to the engine beside the device
        mask, so that the plan is chosen without a device -> host read-back"""
        self._ensure_text()
        dev = self.device
        batch = self.tokenizer(prompt, max_length=self.tokenizer.model_max_length, padding=True, truncation=True,
                               return_tensors="pt")
        ids, am = batch.input_ids.to(dev), batch.attention_mask.to(dev)
        with torch.no_grad():
            pe = self.text_encoder(input_ids=ids, attention_mask=am)[0]
        pe = pe.repeat_interleave(num_samples_per_prompt, 0)
        am = am.repeat_interleave(num_samples_per_prompt, 0)
        ub = self.tokenizer([""] * len(prompt), max_length=pe.shape[1], padding="max_length", truncation=True,
                            return_tensors="pt")
        uids, uam = ub.input_ids.to(dev), ub.attention_mask.to(dev)
        with torch.no_grad():
            ne = self.text_encoder(input_ids=uids, attention_mask=uam)[0]
        ne = ne.repeat_interleave(num_samples_per_prompt, 0)
        uam = uam.repeat_interleave(num_samples_per_prompt, 0)
        host = torch.cat([ub.attention_mask.repeat_interleave(num_samples_per_prompt, 0),
                          batch.attention_mask.repeat_interleave(num_samples_per_prompt, 0)]) == 1      # tokenizer outputs: CPU tensors
        return torch.cat([ne, pe]), (torch.cat([uam, am]) == 1).to(dev), host.cpu()

    # ---- latents -----------------------------------------------------------------------------
    def prepare_latents(self, batch_size, inference_scheduler, num_channels_latents, dtype, device):
        """models.py:259-264 (global torch generator on `device`, like the reference)."""
        shape = (batch_size, num_channels_latents, 256, 16)
        latents = torch.randn(shape, device=device, dtype=dtype)
        return latents * inference_scheduler.init_noise_sigma

    # ---- the hot path ------------------------------------------------------------------------
    def _pad_text(self, embeds, mask, mask_host=None):
        """Static plan shapes: pad L up to a bucket with masked tokens.  Numerically exact: a masked
        key's weight is exp(-10000) == 0 in fp32 (SURVEY.md section 4 differential check)."""
        L = embeds.shape[1]
        if not self.bucket_text_len:
            return embeds, mask, mask_host
        tgt = next((b for b in _TEXT_BUCKETS if b >= L), L)
        if tgt == L:
            return embeds, mask, mask_host
        pe = torch.zeros((embeds.shape[0], tgt, embeds.shape[2]), device=embeds.device, dtype=embeds.dtype)
        pe[:, :L] = embeds
        pm = torch.zeros((mask.shape[0], tgt), device=mask.device, dtype=torch.bool)
        pm[:, :L] = mask
        if mask_host is not None:
            ph = torch.zeros((mask.shape[0], tgt), dtype=torch.bool)
            ph[:, :L] = mask_host
            mask_host = ph
        return pe, pm, mask_host

    def _denoise(self, prompt_embeds, boolean_prompt_mask, inference_scheduler, num_steps, guidance_scale, latents, noise, seed,
                 sample_offset, mask_host=None, **extra_conditions):
        """the shared body of the loop of models.py:224-249 / mustango/models.py:563-598: seed derivation, text bucketing, scheduler
        tables, one engine call (`extra_conditions`: the Music UNet's beat / chord streams)"""
        cfg_on = guidance_scale > 1.0
        B = prompt_embeds.shape[0] // 2 if cfg_on else prompt_embeds.shape[0]
        inference_scheduler.set_timesteps(num_steps, device=self.device)
        timesteps = inference_scheduler.timesteps
        if latents is None:
            latents = self.prepare_latents(B, inference_scheduler, self.unet.config.in_channels, torch.float32, self.device)
        latents = latents.to(self.device, torch.float32).contiguous().clone()
        if boolean_prompt_mask is None:
            boolean_prompt_mask = torch.ones(prompt_embeds.shape[:2], dtype=torch.bool, device=prompt_embeds.device)
        # (the mask is NOT moved here: a host mask reaches the engine as a host pointer too, which spares the call its only host sync)
        pe, pm, mask_host = self._pad_text(prompt_embeds.to(self.device), boolean_prompt_mask, mask_host)
        c = inference_scheduler.config
        if seed is None:
            # the reference draws step noise from torch's global generator (randn_tensor in scheduler.step): derive the
            # Philox key from that generator, so torch.manual_seed() fixes the whole trajectory and calls differ
            seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if self.seed is None else (int(self.seed) << 20) + self._calls
        self._calls += 1
        self.engine.denoise(latents, pe, pm, timesteps.cpu().numpy(), inference_scheduler.coef_table(), guidance_scale,
                            prediction_type=c.prediction_type, rule=inference_scheduler.rule, clip_sample=c.clip_sample,
                            clip_sample_range=getattr(c, "clip_sample_range", 1.0), noise=noise, seed=seed,
                            sample_offset=sample_offset, use_graph=self.use_graph, prompt_mask_host=mask_host, **extra_conditions)
        return latents

    @torch.no_grad()
    def inference_from_embeddings(self, prompt_embeds, boolean_prompt_mask, inference_scheduler, num_steps=20,
                                  guidance_scale=3, latents=None, noise=None, seed=None, sample_offset=0, mask_host=None):
        """Loop of models.py:224-249 given the encoder outputs ([uncond; cond] when guidance > 1).  `mask_host`: optional CPU copy of
        a device-resident `boolean_prompt_mask` (see Engine.denoise)."""
        return self._denoise(prompt_embeds, boolean_prompt_mask, inference_scheduler, num_steps, guidance_scale, latents, noise, seed,
                             sample_offset, mask_host=mask_host)

    @torch.no_grad()
    def inference(self, prompt, inference_scheduler, num_steps=20, guidance_scale=3, num_samples_per_prompt=1,
                  disable_progress=True):
        """models.py:210-257 (same signature, same return: latents [B*S, 8, 256, 16])."""
        host = None
        if guidance_scale > 1.0:
            pe, pm, host = self._encode_text_classifier_free(prompt, num_samples_per_prompt)
        else:
            pe, pm = self.encode_text(prompt)
            pe = pe.repeat_interleave(num_samples_per_prompt, 0)
            pm = pm.repeat_interleave(num_samples_per_prompt, 0)
        return self.inference_from_embeddings(pe.float(), pm, inference_scheduler, num_steps, guidance_scale, mask_host=host)


class MusicAudioDiffusion(AudioDiffusion):
    """Inference side of Mustango's `MusicAudioDiffusion` (mustango/models.py:312-740) on the engine: the UNet is
    UNet2DConditionModelMusic (mustango/configs/music_diffusion_model_config.json -- every cross-attention site attends to the
    text, then the beat, then the chord embeddings) and the loop of mustango/models.py:563-598 runs as the same captured
    hipGraph step with three condition tensors bound.

    What stays with the caller, exactly like the T5 encoder of `AudioDiffusion`: the symbolic front-end that turns beat / chord
    annotations into embeddings -- `beat_tokenizer`, `chord_tokenizer`, `Beat_Embedding`, `Chord_Embedding`,
    `Fundamental_Music_Embedding`, `Music_PositionalEncoding` (mustango/layers/layers.py, mustango/models.py:376-388,433-467):
    small trainable torch modules whose outputs ([B, 50, 1024] beats, [B, 20, 1024] chords and their masks,
    `encode_beats_classifier_free` / `encode_chords_classifier_free` ordered [uncond; cond]) are this class's inputs."""

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        if not self.unet_config.get("music"):
            raise ValueError("MusicAudioDiffusion needs a UNet config with *Music blocks (mustango/configs/music_diffusion_model_config.json)")

    @torch.no_grad()
    def inference_from_embeddings(self, prompt_embeds, boolean_prompt_mask, inference_scheduler, num_steps=20, guidance_scale=3,
                                  latents=None, noise=None, seed=None, sample_offset=0, *, encoded_beats=None, beat_mask=None,
                                  encoded_chords=None, chord_mask=None):
        """mustango/models.py:540-598 given the three encoder outputs ([uncond; cond] when guidance > 1)."""
        if encoded_beats is None or encoded_chords is None:
            raise ValueError("encoded_beats and encoded_chords are required (mustango/models.py:548-550)")
        return self._denoise(prompt_embeds, boolean_prompt_mask, inference_scheduler, num_steps, guidance_scale, latents, noise, seed,
                             sample_offset, beat_embeds=encoded_beats, beat_mask=beat_mask, chord_embeds=encoded_chords,
                             chord_mask=chord_mask)

    def inference(self, *a, **k):
        raise NotImplementedError("strings / beat and chord annotations are encoded by the caller's Mustango front-end modules; "
                                  "pass their outputs to inference_from_embeddings()")
