"""Thin Python handle over the C-ABI engine.  PyTorch is plumbing only: device memory, streams."""
import ctypes as C
import os
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from . import weights as W

#: configs/diffusion_model_config.json of the reference (FLAN-T5-large Tango / Tango-full / Tango2)
UNET_CONFIG_LARGE = dict(
    in_channels=8, out_channels=8, block_out_channels=[320, 640, 1280, 1280],
    attention_head_dim=[5, 10, 20, 20], layers_per_block=2, cross_attention_dim=1024,
    down_block_types=["CrossAttnDownBlock2D"] * 3 + ["DownBlock2D"],
    up_block_types=["UpBlock2D"] + ["CrossAttnUpBlock2D"] * 3,
    norm_num_groups=32, norm_eps=1e-5, flip_sin_to_cos=True, freq_shift=0,
)
UNET_CONFIG_XL = dict(UNET_CONFIG_LARGE, cross_attention_dim=2048)
#: mustango/configs/music_diffusion_model_config.json (Mustango's UNet2DConditionModelMusic)
UNET_CONFIG_MUSIC = dict(UNET_CONFIG_LARGE, down_block_types=["CrossAttnDownBlock2DMusic"] * 3 + ["DownBlock2D"],
                         up_block_types=["UpBlock2D"] + ["CrossAttnUpBlock2DMusic"] * 3)
#: mustango/configs/vae_config.json ddconfig + scale_factor
VAE_CONFIG = dict(ch=128, ch_mult=[1, 2, 4], num_res_blocks=2, z_channels=8, out_ch=1, embed_dim=8,
                  scale_factor=0.9227914214134216)
#: audioldm/hifigan/utilities.py:9-39
HIFIGAN_CONFIG = dict(upsample_rates=[5, 4, 2, 2, 2], upsample_kernel_sizes=[16, 16, 8, 4, 4],
                      upsample_initial_channel=1024, resblock_kernel_sizes=[3, 7, 11],
                      resblock_dilation_sizes=[[1, 3, 5]] * 3, num_mels=64)


def normalize_unet_config(cfg: dict) -> dict:
    out = dict(UNET_CONFIG_LARGE)
    for k in out:
        if k in cfg:
            out[k] = cfg[k]
    if isinstance(out["attention_head_dim"], int):
        out["attention_head_dim"] = [out["attention_head_dim"]] * len(out["block_out_channels"])
    # the engine derives the up path from the down path (unet_2d_condition.py:330-375 builds them independently):
    # refuse configurations where up_block_types is not the mirror image instead of silently assuming it
    down, up = list(out["down_block_types"]), list(out["up_block_types"])
    # Mustango's music UNet (mustango/configs/music_diffusion_model_config.json): the same topology with *Music cross-attention
    # blocks (three transformers per site); a config is either all-Music or all-plain at its cross-attention sites
    music = any(t.endswith("Music") for t in down + up)
    xd, xu = ("CrossAttnDownBlock2DMusic", "CrossAttnUpBlock2DMusic") if music else ("CrossAttnDownBlock2D", "CrossAttnUpBlock2D")
    known_d, known_u = {xd, "DownBlock2D"}, {xu, "UpBlock2D"}
    if set(down) - known_d or set(up) - known_u:
        raise ValueError("unsupported UNet block types %s / %s" % (down, up))
    mirror = [xu if d == xd else "UpBlock2D" for d in reversed(down)]
    if up != mirror or len(down) != len(out["block_out_channels"]):
        raise ValueError("up_block_types %s must mirror down_block_types %s" % (up, down))
    mid = cfg.get("mid_block_type")
    if mid is not None and mid != ("UNetMidBlock2DCrossAttnMusic" if music else "UNetMidBlock2DCrossAttn"):
        raise ValueError("unsupported mid_block_type %s" % mid)
    out["music"] = music
    return out


def _stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Engine:
    """One engine handle = one (device, model) pair; not re-entrant (include/tango_engine.h)."""

    def __init__(self, unet: Optional[dict] = None, vae: Optional[dict] = None, hifigan: Optional[dict] = None,
                 dtype: str = "fp16", device="cuda:0", t5: Optional[dict] = None, vae_encoder: bool = False,
                 stft: Optional[dict] = None, attn_fp8: bool = False):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("tango_amd.Engine needs a HIP device (no CPU fallback)")
        self.device = torch.device(device)
        self.dtype = dtype
        self.unet_cfg = normalize_unet_config(unet) if unet is not None else None
        self.vae_cfg = dict(vae) if vae is not None else None
        self.hifigan_cfg = dict(hifigan) if hifigan is not None else None
        self.t5_cfg = dict(t5) if t5 is not None else None
        self.vae_encoder = bool(vae_encoder) and vae is not None
        self.stft_cfg = dict(stft) if stft is not None else None
        self.attn_fp8 = int(attn_fp8)      # False / True, or 2: the MX instruction (128 keys per MFMA) where the sequence allows
        if self.attn_fp8 and dtype in ("fp32", "float32", "f32"):
            raise ValueError("attn_fp8 (P.V on the fp8 MFMA) needs a 16-bit engine dtype")
        c = _lib.TangoConfig()
        c.dtype = _lib.DTYPES[dtype]
        c.latent_h, c.latent_w = 256, 16
        if self.unet_cfg is not None:
            u = self.unet_cfg
            ch = u["block_out_channels"]
            c.unet_levels = len(ch)
            for i, v in enumerate(ch):
                c.unet_channels[i] = v
                c.unet_heads[i] = u["attention_head_dim"][i]
                c.unet_cross_attn[i] = 1 if u["down_block_types"][i].startswith("CrossAttnDownBlock2D") else 0
            c.unet_music = 1 if u.get("music") else 0
            c.unet_attn_fp8 = int(attn_fp8)
            c.unet_layers_per_block = u["layers_per_block"]
            c.unet_in_channels = u["in_channels"]
            c.unet_out_channels = u["out_channels"]
            c.unet_cross_dim = u["cross_attention_dim"]
            c.unet_groups = u["norm_num_groups"]
            c.unet_eps = u["norm_eps"]
            c.unet_flip_sin_to_cos = 1 if u["flip_sin_to_cos"] else 0
            c.unet_freq_shift = float(u["freq_shift"])
        if self.vae_cfg is not None:
            v = self.vae_cfg
            c.vae_levels = len(v["ch_mult"])
            c.vae_ch = v["ch"]
            for i, m in enumerate(v["ch_mult"]):
                c.vae_ch_mult[i] = m
            c.vae_num_res_blocks = v["num_res_blocks"]
            c.vae_z_channels = v["z_channels"]
            c.vae_embed_dim = v.get("embed_dim", 8)
            c.vae_out_ch = v["out_ch"]
            c.vae_scale_factor = v["scale_factor"]
            c.vae_encoder = 1 if self.vae_encoder else 0
            c.vae_in_channels = v.get("in_channels", 1)
        if self.hifigan_cfg is not None:
            h = self.hifigan_cfg
            c.voc_n_ups = len(h["upsample_rates"])
            for i, (r, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
                c.voc_rates[i] = r
                c.voc_kernels[i] = k
            c.voc_initial_channel = h["upsample_initial_channel"]
            c.voc_num_mels = h["num_mels"]
            c.voc_n_resblocks = len(h["resblock_kernel_sizes"])
            for j, k in enumerate(h["resblock_kernel_sizes"]):
                c.voc_res_kernels[j] = k
                for m, d in enumerate(h["resblock_dilation_sizes"][j]):
                    c.voc_res_dilations[j][m] = d
        if self.t5_cfg is not None:
            t = self.t5_cfg
            c.t5_layers, c.t5_d_model, c.t5_d_kv, c.t5_heads = t["num_layers"], t["d_model"], t["d_kv"], t["num_heads"]
            c.t5_d_ff, c.t5_vocab = t["d_ff"], t["vocab_size"]
            c.t5_rel_buckets = t.get("relative_attention_num_buckets", 32)
            c.t5_rel_max_distance = t.get("relative_attention_max_distance", 128)
            c.t5_eps = t.get("layer_norm_epsilon", 1e-6)
        if self.stft_cfg is not None:
            c.stft_filter_length, c.stft_hop_length = self.stft_cfg["filter_length"], self.stft_cfg["hop_length"]
            c.stft_n_mel = self.stft_cfg["n_mel_channels"]
        self._cfg = c
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.tango_engine_create(C.byref(c), C.byref(self._h)), "tango_engine_create")
        self._finalized = False

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self.lib.tango_engine_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- weights -------------------------------------------------------------------------
    def weight_names(self):
        n = self.lib.tango_engine_num_weights(self._h)
        return [self.lib.tango_engine_weight_name(self._h, i).decode() for i in range(n)]

    def expected_shapes(self) -> Dict[str, tuple]:
        out = {}
        if self.unet_cfg is not None:
            out.update(W.unet_param_shapes(self.unet_cfg, "unet."))
        if self.vae_cfg is not None:
            out.update(W.vae_decoder_param_shapes(self.vae_cfg))
            if self.vae_encoder:
                out.update(W.vae_encoder_param_shapes(self.vae_cfg))
        if self.hifigan_cfg is not None:
            out.update(W.hifigan_param_shapes(self.hifigan_cfg))
        if self.t5_cfg is not None:
            out.update(W.t5_encoder_param_shapes(self.t5_cfg))
        if self.stft_cfg is not None:
            out.update(W.stft_param_shapes(self.stft_cfg))
        return out

    def set_weight(self, name: str, tensor: torch.Tensor):
        t = tensor.detach().to(device=self.device, dtype=torch.float32).contiguous()
        shape = (C.c_int64 * t.dim())(*t.shape)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.tango_engine_set_weight(self._h, name.encode(), C.c_void_p(t.data_ptr()), shape, t.dim()),
                       "set_weight(%s)" % name)

    def load_state_dict(self, sd: Dict[str, torch.Tensor], prefix: str = "", strict: bool = True):
        """Ingest reference-named tensors (pytorch_model_main.bin / pytorch_model_vae.bin keys).
        Keys the engine does not need (text_encoder.*, encoder.*, quant_conv.*) are ignored."""
        need = set(self.weight_names())
        for k in list(need):
            src = prefix + k
            if src in sd:
                self.set_weight(k, sd[src])
                need.discard(k)
        if strict and need:
            raise RuntimeError("Missing key(s) in state_dict: %s ..." % sorted(need)[:4])
        return sorted(need)

    def load_synthetic(self, seed: int = 1234):
        """Seeded synthetic weights (tango_amd.weights.synth_tensor), streamed tensor by tensor."""
        shapes = self.expected_shapes()
        for k in self.weight_names():
            self.set_weight(k, W.synth_tensor(k, shapes[k], seed))
        self.finalize()

    def finalize(self):
        _lib.check(self.lib.tango_engine_finalize_weights(self._h), "finalize_weights")
        self._finalized = True

    # ---- compute -------------------------------------------------------------------------
    def _f32(self, t):
        return t.detach().to(device=self.device, dtype=torch.float32).contiguous()

    def _u8(self, m):
        return m.to(self.device).to(torch.uint8).contiguous() if m is not None else None

    def _check_latents(self, what, x):
        """the UNet plans are built for ONE latent size: [*, in_channels, latent_h, latent_w]"""
        want = (self.unet_cfg["in_channels"], self._cfg.latent_h, self._cfg.latent_w)
        if x.dim() != 4 or tuple(x.shape[1:]) != want:
            raise ValueError("%s must be [B, %d, %d, %d], got %s" % ((what,) + want + (tuple(x.shape),)))

    def _check_cond(self, what, emb, mask, rows):
        """a condition stream [rows, L, cross_attention_dim] and its optional mask [rows, L]: the engine reads exactly that many
        bytes through the raw pointers, so a mis-shaped tensor must fail here, not as a device out-of-bounds read"""
        d = self.unet_cfg["cross_attention_dim"]
        if emb.dim() != 3 or emb.shape[0] != rows or emb.shape[2] != d or emb.shape[1] < 1:
            raise ValueError("%s must be [%d, L, %d], got %s" % (what, rows, d, tuple(emb.shape)))
        if mask is not None and tuple(mask.shape) != tuple(emb.shape[:2]):
            raise ValueError("%s mask must be %s, got %s" % (what, tuple(emb.shape[:2]), tuple(mask.shape)))

    def unet_forward(self, sample, timestep, encoder_hidden_states, encoder_attention_mask=None, beat_features=None,
                     chord_features=None, beat_attention_mask=None, chord_attention_mask=None):
        """UNet2DConditionModel.forward; with beat / chord features, UNet2DConditionModelMusic.forward (Music configs only)"""
        x = self._f32(sample)
        enc = self._f32(encoder_hidden_states)
        self._check_latents("sample", x)
        B2, L = x.shape[0], enc.shape[1] if enc.dim() == 3 else 0
        mask = self._u8(encoder_attention_mask)
        self._check_cond("encoder_hidden_states", enc, mask, B2)
        out = torch.empty_like(x)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731
        music = bool(self.unet_cfg.get("music"))
        given = (beat_features is not None, chord_features is not None)
        if (music and not all(given)) or (not music and any(given)):
            raise ValueError("beat_features / chord_features are required by (and only by) a Music UNet config")
        with torch.cuda.device(self.device):
            if music:
                beat, chord = self._f32(beat_features), self._f32(chord_features)
                bm, cm = self._u8(beat_attention_mask), self._u8(chord_attention_mask)
                self._check_cond("beat_features", beat, bm, B2)
                self._check_cond("chord_features", chord, cm, B2)
                _lib.check(self.lib.tango_engine_unet_forward_music(
                    self._h, p(x), int(timestep), p(enc), p(mask), p(beat), p(bm), p(chord), p(cm), p(out), B2, L, beat.shape[1],
                    chord.shape[1], _stream_ptr()), "unet_forward_music")
            else:
                _lib.check(self.lib.tango_engine_unet_forward(self._h, p(x), int(timestep), p(enc), p(mask), p(out), B2, L,
                                                              _stream_ptr()), "unet_forward")
        return out

    def denoise(self, latents, prompt_embeds, prompt_mask, timesteps, coef, guidance_scale, prediction_type="v_prediction",
                rule="ddpm", clip_sample=False, clip_sample_range=1.0, noise=None, seed=0, sample_offset=0, use_graph=True,
                beat_embeds=None, beat_mask=None, chord_embeds=None, chord_mask=None, prompt_mask_host=None):
        """In-place denoise of `latents` [B,8,256,16] (fp32 cuda).  `timesteps` int64 [N] and `coef`
        float32 [N,8] are host tables from tango_amd.scheduler.  `prompt_mask_host`: the caller's CPU copy of a `prompt_mask`
        that already lives on the device (the tokenizer's attention mask): the engine then picks its plan from it instead of
        reading the device mask back, i.e. the call does not synchronise the host."""
        assert latents.is_cuda and latents.dtype == torch.float32 and latents.is_contiguous()
        self._check_latents("latents", latents)
        enc = self._f32(prompt_embeds)
        mask = prompt_mask.to(self.device).to(torch.uint8).contiguous() if prompt_mask is not None else None
        # a mask that is still on the host (tokenizer output) also travels as a host pointer: the engine then picks its plan without
        # reading the device copy back (no host sync inside the call)
        mask_host = prompt_mask.to(torch.uint8).contiguous() if prompt_mask is not None and not prompt_mask.is_cuda else None
        if mask_host is None and prompt_mask_host is not None and prompt_mask is not None:
            if prompt_mask_host.is_cuda or tuple(prompt_mask_host.shape) != tuple(prompt_mask.shape):
                raise ValueError("prompt_mask_host must be a CPU tensor of prompt_mask's shape %s" % (tuple(prompt_mask.shape),))
            mask_host = prompt_mask_host.to(torch.uint8).contiguous()
            # ONE source of truth: the plan (single-key prefix, CFG-shared prefix) is picked from the host copy and the
            # attention kernels read the device mask, so the device mask is re-derived from the host copy here (a 2B x L byte upload,
            # asynchronous: still no host sync) instead of trusting that the caller's two copies are identical.
            # TANGO_DEBUG_MASK=1 additionally compares the caller's device mask with it (one D2H + sync) and fails on a difference.
            if os.environ.get("TANGO_DEBUG_MASK") and not torch.equal(mask.cpu(), mask_host):
                raise ValueError("prompt_mask_host differs from the device prompt_mask")
            mask = mask_host.to(self.device, non_blocking=True)
        rows = 2 * latents.shape[0] if guidance_scale > 1.0 else latents.shape[0]
        self._check_cond("prompt_embeds", enc, mask, rows)
        ts = np.ascontiguousarray(np.asarray(timesteps, dtype=np.int64))
        cf = np.ascontiguousarray(np.asarray(coef, dtype=np.float32))
        assert cf.shape == (len(ts), 8)
        if noise is not None:
            noise = self._f32(noise)
            if tuple(noise.shape) != (len(ts),) + tuple(latents.shape):
                raise ValueError("noise must be [num_steps, %s], got %s" % (", ".join(map(str, latents.shape)), tuple(noise.shape)))
        a = _lib.DenoiseArgs()
        a.latents = latents.data_ptr()
        a.prompt_embeds = enc.data_ptr()
        a.prompt_mask = mask.data_ptr() if mask is not None else None
        a.prompt_mask_host = mask_host.data_ptr() if mask_host is not None else None
        a.batch = latents.shape[0]
        a.text_len = enc.shape[1]
        a.num_steps = len(ts)
        a.timesteps = ts.ctypes.data
        a.coef = cf.ctypes.data
        a.guidance_scale = float(guidance_scale)
        a.prediction_type = _lib.PRED[prediction_type]
        a.rule = _lib.RULE[rule]
        a.clip_sample = 1 if clip_sample else 0
        a.clip_sample_range = float(clip_sample_range)
        a.noise = noise.data_ptr() if noise is not None else None
        a.seed = int(seed)
        a.sample_offset = int(sample_offset)
        a.use_graph = 1 if use_graph else 0
        keep = []
        if self.unet_cfg.get("music"):
            if beat_embeds is None or chord_embeds is None:
                raise ValueError("a Music UNet needs beat_embeds and chord_embeds")
            be, ce = self._f32(beat_embeds), self._f32(chord_embeds)
            bm, cm = self._u8(beat_mask), self._u8(chord_mask)
            keep = [be, ce, bm, cm]
            self._check_cond("beat_embeds", be, bm, rows)
            self._check_cond("chord_embeds", ce, cm, rows)
            a.beat_embeds, a.beat_len, a.beat_mask = be.data_ptr(), be.shape[1], bm.data_ptr() if bm is not None else None
            a.chord_embeds, a.chord_len, a.chord_mask = ce.data_ptr(), ce.shape[1], cm.data_ptr() if cm is not None else None
        with torch.cuda.device(self.device):
            _lib.check(self.lib.tango_engine_denoise(self._h, C.byref(a), _stream_ptr()), "denoise")
        return latents

    def profile_unet(self, batch2: int, text_len: int):
        """per-op timing of one eager UNet step: list of (label, ms, gflop)"""
        buf = C.create_string_buffer(1 << 20)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.tango_engine_profile_unet(self._h, batch2, text_len, buf, len(buf), _stream_ptr()), "profile_unet")
        rows = []
        for line in buf.value.decode().splitlines():
            lab, ms, gf = line.rsplit("\t", 2)
            rows.append((lab, float(ms), float(gf)))
        return rows

    def _profile(self, fn, what, *args):
        buf = C.create_string_buffer(1 << 20)
        with torch.cuda.device(self.device):
            _lib.check(fn(self._h, *args, buf, len(buf), _stream_ptr()), what)
        rows = []
        for line in buf.value.decode().splitlines():
            lab, ms, gf = line.rsplit("\t", 2)
            rows.append((lab, float(ms), float(gf)))
        return rows

    def profile_vae(self, batch: int):
        """per-op timing of the mel-VAE decoder plan for `batch` latents: list of (label, ms, gflop)"""
        return self._profile(self.lib.tango_engine_profile_vae, "profile_vae", int(batch))

    def profile_vocoder(self, batch: int, frames: int = 1024):
        """per-op timing of the HiFi-GAN plan for `batch` mels of `frames` frames: list of (label, ms, gflop)"""
        return self._profile(self.lib.tango_engine_profile_vocoder, "profile_vocoder", int(batch), int(frames))

    def set_plan_budget(self, nbytes: int):
        """byte budget of the plan cache (workspace slabs kept alive per call shape); least recently used plans are freed first"""
        _lib.check(self.lib.tango_engine_set_plan_budget(self._h, int(nbytes)), "set_plan_budget")

    def drop_plans(self):
        """free every cached plan (workspace slab + launch program); the next call of each shape rebuilds its own"""
        _lib.check(self.lib.tango_engine_drop_plans(self._h), "drop_plans")

    def plan_stats(self):
        """(bytes of plan workspace alive, number of cached plans)"""
        b, n = C.c_uint64(), C.c_int()
        _lib.check(self.lib.tango_engine_plan_stats(self._h, C.byref(b), C.byref(n)), "plan_stats")
        return int(b.value), int(n.value)

    def last_step_gflop(self):
        """GFLOP one launch of the last denoise call's step program executes (None before the first denoise call)"""
        g = C.c_double()
        if self.lib.tango_engine_last_step_gflop(self._h, C.byref(g)) != 0:
            return None
        return float(g.value)

    def last_denoise_ms(self):
        tot, per = C.c_float(), C.c_float()
        _lib.check(self.lib.tango_engine_last_denoise_ms(self._h, C.byref(tot), C.byref(per)), "last_denoise_ms")
        return tot.value, per.value

    def encode_text(self, input_ids, attention_mask=None):
        """FLAN-T5 encoder forward: int64 ids [B, L] (+ 0/1 mask) -> fp32 last_hidden_state [B, L, d_model] on the device."""
        ids = input_ids.detach().to(device=self.device, dtype=torch.int64).contiguous()
        if ids.dim() != 2 or ids.numel() == 0:
            raise ValueError("encode_text: input_ids must be [B, L], got %s" % (tuple(ids.shape),))
        B, L = ids.shape
        lo, hi = int(ids.min()), int(ids.max())           # one sync per prompt batch; nn.Embedding device-asserts on the same condition
        if lo < 0 or hi >= self.t5_cfg["vocab_size"]:
            raise IndexError("encode_text: token id %d outside the embedding table [0, %d)" % (lo if lo < 0 else hi, self.t5_cfg["vocab_size"]))
        m = attention_mask.detach().to(self.device).to(torch.uint8).contiguous() if attention_mask is not None else None
        if m is not None and tuple(m.shape) != (B, L):
            raise ValueError("encode_text: attention_mask must be %s, got %s" % ((B, L), tuple(m.shape)))
        out = torch.empty((B, L, self.t5_cfg["d_model"]), device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.tango_engine_encode_text(self._h, C.c_void_p(ids.data_ptr()), C.c_void_p(m.data_ptr()) if m is not None else None,
                                                         C.c_void_p(out.data_ptr()), B, L, _stream_ptr()), "encode_text")
        return out

    def mel_spectrogram(self, wav, want_log_magnitudes=True, want_energy=True):
        """TacotronSTFT.mel_spectrogram on the engine: wav [B, N] fp32 in [-1, 1] -> (log-mel [B, n_mel, T],
        log-magnitudes [B, n_fft/2+1, T] or None, energy [B, T] or None), T = 1 + N // hop"""
        if self.stft_cfg is None:
            raise RuntimeError("Engine was created without an stft config")
        y = self._f32(wav)
        if y.dim() != 2:
            raise ValueError("mel_spectrogram: wav must be [B, n_samples], got %s" % (tuple(y.shape),))
        B, N = y.shape
        T = self.lib.tango_engine_mel_frames(self._h, N)
        cutoff = self.stft_cfg["filter_length"] // 2 + 1
        mel = torch.empty((B, self.stft_cfg["n_mel_channels"], T), device=self.device, dtype=torch.float32)
        lm = torch.empty((B, cutoff, T), device=self.device, dtype=torch.float32) if want_log_magnitudes else None
        en = torch.empty((B, T), device=self.device, dtype=torch.float32) if want_energy else None
        nf = C.c_int()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.tango_engine_mel_spectrogram(
                self._h, C.c_void_p(y.data_ptr()), C.c_void_p(mel.data_ptr()), C.c_void_p(lm.data_ptr()) if lm is not None else None,
                C.c_void_p(en.data_ptr()) if en is not None else None, B, N, C.byref(nf), _stream_ptr()), "mel_spectrogram")
        assert nf.value == T
        return mel, lm, en

    def vae_decode(self, latents):
        z = self._f32(latents)
        B = z.shape[0]
        nl = len(self.vae_cfg["ch_mult"])
        want = (self.vae_cfg.get("embed_dim", 8), self._cfg.latent_h, self._cfg.latent_w)   # the plan's fixed input size
        if z.dim() != 4 or tuple(z.shape[1:]) != want:
            raise ValueError("vae_decode: latents must be [B, %d, %d, %d], got %s" % (want + (tuple(z.shape),)))
        mel = torch.empty((B, self.vae_cfg["out_ch"], z.shape[2] << (nl - 1), z.shape[3] << (nl - 1)),
                          device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.tango_engine_vae_decode(self._h, C.c_void_p(z.data_ptr()), C.c_void_p(mel.data_ptr()), B,
                                                        _stream_ptr()), "vae_decode")
        return mel

    def vae_encode(self, mel):
        """mel [B, 1, 4H, 4W] fp32 -> moments [B, 2*embed_dim, H, W] fp32 (= [mean | logvar]; autoencoder.py:52-58 up to quant_conv)"""
        if not self.vae_encoder:
            raise RuntimeError("Engine was created without vae_encoder=True")
        x = self._f32(mel)
        B = x.shape[0]
        nl = len(self.vae_cfg["ch_mult"])
        # the engine's encoder plan is built for ONE spatial size, (latent_h, latent_w) << (levels - 1) = 1024 x 64 mel bins
        # (the C ABI sees a bare pointer and copies exactly that many bytes): refuse anything else instead of reading past
        # `mel` / writing past `moments`.  The reference Encoder accepts any H, W; Tango only ever feeds this one.
        want = (self.vae_cfg.get("in_channels", 1), self._cfg.latent_h << (nl - 1), self._cfg.latent_w << (nl - 1))
        if x.dim() != 4 or tuple(x.shape[1:]) != want:
            raise ValueError("vae_encode: mel must be [B, %d, %d, %d], got %s" % (want + (tuple(x.shape),)))
        mom = torch.empty((B, 2 * self.vae_cfg.get("embed_dim", 8), x.shape[2] >> (nl - 1), x.shape[3] >> (nl - 1)),
                          device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.tango_engine_vae_encode(self._h, C.c_void_p(x.data_ptr()), C.c_void_p(mom.data_ptr()), B,
                                                        _stream_ptr()), "vae_encode")
        return mom

    def vocoder_samples(self, frames: int) -> int:
        return self.lib.tango_engine_vocoder_samples(self._h, frames)

    def vocode(self, mel):
        """mel [B,1,T,num_mels] fp32 -> int16 cuda tensor [B, samples]"""
        m = self._f32(mel)
        nm = self.hifigan_cfg.get("num_mels", 64)
        if m.dim() != 4 or m.shape[1] != 1 or m.shape[3] != nm or m.shape[2] < 1:
            raise ValueError("vocode: mel must be [B, 1, T, %d], got %s" % (nm, tuple(m.shape)))
        B, T = m.shape[0], m.shape[2]
        n = self.vocoder_samples(T)
        wav = torch.empty((B, n), device=self.device, dtype=torch.int16)
        ns = C.c_int()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.tango_engine_vocode(self._h, C.c_void_p(m.data_ptr()), C.c_void_p(wav.data_ptr()), B, T,
                                                    C.byref(ns), _stream_ptr()), "vocode")
        assert ns.value == n
        return wav
