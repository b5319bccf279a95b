"""CPU oracle for the Tango text-to-audio hot path.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module.  `tango_amd/` never does: the product path fails loudly when the HIP library is missing.

What it is: a functional, fp32, plain-`torch`-on-CPU restatement of the reference algorithm for the
path   prompt-embeddings -> CFG denoise loop (UNet + DDPM/DDIM step) -> mel-VAE decode -> HiFi-GAN
-> int16.  Every function takes the reference's own `state_dict` tensors (reference key names) and
cites the reference file:line it follows (paths relative to /root/reference; "fork" =
mustango/diffusers/src/diffusers, the in-tree diffusers 0.15.0.dev0 the reference was developed on;
the pinned pip wheel diffusers==0.18.2, requirements.txt:7, is absent from the tree).

Parity pinning (see tests/test_oracle_*.py and tests/golden/):
  * schedulers: the fork's own known-answer tests (tests/schedulers/test_scheduler_ddpm.py:62-131,
    test_scheduler_ddim.py:46-54,106-140) are re-run against this restatement;
  * UNet blocks / layers: the fork's KATs (tests/test_unet_2d_blocks.py, tests/test_layers_utils.py)
    via fixtures produced by oracle/make_golden.py from the *imported reference modules*;
  * full UNet (incl. the encoder_attention_mask path the fork's tests never exercise), VAE decoder,
    HiFi-GAN, int16 cast: differential fixtures generated from the imported reference modules
    (oracle/ref_import.py) -- "parity unpinned by the reference's own tests", pinned by outputs of
    the reference itself run in the build container.
"""
import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

# ----------------------------------------------------------------------------------------------
# configs
# ----------------------------------------------------------------------------------------------

#: configs/diffusion_model_config.json (FLAN-T5-large); `attention_head_dim` is the HEAD COUNT
#: (fork models/unet_2d_blocks.py:983-986), head_dim is always C/heads = 64.
UNET_CONFIG_LARGE = dict(
    in_channels=8, out_channels=8, block_out_channels=[320, 640, 1280, 1280],
    attention_head_dim=[5, 10, 20, 20], layers_per_block=2, cross_attention_dim=1024,
    down_block_types=["CrossAttnDownBlock2D"] * 3 + ["DownBlock2D"],
    up_block_types=["UpBlock2D"] + ["CrossAttnUpBlock2D"] * 3,
    norm_num_groups=32, norm_eps=1e-5, flip_sin_to_cos=True, freq_shift=0,
)
#: configs/diffusion_model_config_xl.json differs only in cross_attention_dim
UNET_CONFIG_XL = dict(UNET_CONFIG_LARGE, cross_attention_dim=2048)

#: A width-reduced config with the same topology (head_dim stays 64) for fast CPU tests.
UNET_CONFIG_TINY = dict(
    UNET_CONFIG_LARGE, block_out_channels=[64, 128, 256, 256], attention_head_dim=[1, 2, 4, 4],
    cross_attention_dim=96,
)

#: mustango/configs/music_diffusion_model_config.json (Mustango): the Tango UNet with *Music cross-attention blocks
UNET_CONFIG_MUSIC = dict(
    UNET_CONFIG_LARGE,
    down_block_types=["CrossAttnDownBlock2DMusic"] * 3 + ["DownBlock2D"],
    up_block_types=["UpBlock2D"] + ["CrossAttnUpBlock2DMusic"] * 3,
)
UNET_CONFIG_MUSIC_TINY = dict(UNET_CONFIG_MUSIC, block_out_channels=[64, 128, 256, 256], attention_head_dim=[1, 2, 4, 4],
                              cross_attention_dim=96)

#: mustango/configs/vae_config.json == audioldm/utils.py:158-181
VAE_CONFIG = dict(ch=128, ch_mult=[1, 2, 4], num_res_blocks=2, z_channels=8, out_ch=1, embed_dim=8,
                  scale_factor=0.9227914214134216)
VAE_CONFIG_TINY = dict(VAE_CONFIG, ch=32)

#: audioldm/hifigan/utilities.py:9-39 (HIFIGAN_16K_64)
HIFIGAN_CONFIG = dict(upsample_rates=[5, 4, 2, 2, 2], upsample_kernel_sizes=[16, 16, 8, 4, 4],
                      upsample_initial_channel=1024, resblock_kernel_sizes=[3, 7, 11],
                      resblock_dilation_sizes=[[1, 3, 5]] * 3, num_mels=64)
HIFIGAN_CONFIG_TINY = dict(HIFIGAN_CONFIG, upsample_initial_channel=128)

#: stabilityai/stable-diffusion-2-1 scheduler JSON (tango.py:36) -- NOT in the tree; values per
#: SURVEY.md section 0 fact 3.  Config is data: every consumer takes it as an argument.
SD21_SCHEDULER = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                      beta_schedule="scaled_linear", prediction_type="v_prediction",
                      clip_sample=False, variance_type="fixed_small")


def normalize_unet_config(cfg: dict) -> dict:
    out = dict(UNET_CONFIG_LARGE)
    for k in out:
        if k in cfg:
            out[k] = cfg[k]
    return out


# ----------------------------------------------------------------------------------------------
# schedulers
# ----------------------------------------------------------------------------------------------

class DDPMOracle:
    """fork schedulers/scheduling_ddpm.py:123-349 (fp32 tables, bit-exact integer timesteps)."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02,
                 beta_schedule="linear", variance_type="fixed_small", clip_sample=True,
                 prediction_type="epsilon", clip_sample_range=1.0, **_ignored):
        # scheduling_ddpm.py:139-151
        if beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                                        dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(beta_schedule)
        self.alphas = 1.0 - self.betas                       # :157
        self.alphas_cumprod = torch.cumprod(self.alphas, 0)  # :158
        self.one = torch.tensor(1.0)
        self.init_noise_sigma = 1.0                          # :162
        self.num_train_timesteps = num_train_timesteps
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy())
        self.variance_type = variance_type
        self.clip_sample = clip_sample
        self.clip_sample_range = clip_sample_range
        self.prediction_type = prediction_type
        self.order = 1

    def scale_model_input(self, sample, timestep=None):      # :170-182 identity
        return sample

    def set_timesteps(self, n: int):                         # :184-204
        if n > self.num_train_timesteps:
            raise ValueError("num_inference_steps %d > num_train_timesteps %d" % (n, self.num_train_timesteps))
        self.num_inference_steps = n
        ratio = self.num_train_timesteps // n
        self.timesteps = torch.from_numpy((np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64))

    def _prev_t(self, t):
        n = self.num_inference_steps if self.num_inference_steps else self.num_train_timesteps
        return t - self.num_train_timesteps // n

    def variance(self, t):                                   # :206-241 (fixed_small / fixed_large)
        t = int(t)
        prev_t = self._prev_t(t)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        cur_beta = 1 - a_t / a_prev
        var = (1 - a_prev) / (1 - a_t) * cur_beta
        if self.variance_type == "fixed_small":
            var = torch.clamp(var, min=1e-20)
        elif self.variance_type == "fixed_large":
            var = cur_beta
        else:
            raise NotImplementedError(self.variance_type)
        return var

    def coefficients(self, t):
        """the five fp32 scalars of one step: sqrt(abar_t), sqrt(1-abar_t),
        coef(x0), coef(x_t), sqrt(variance) (0 when t == 0)."""
        t = int(t)
        prev_t = self._prev_t(t)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        b_t = 1 - a_t
        b_prev = 1 - a_prev
        cur_a = a_t / a_prev
        cur_b = 1 - cur_a
        c_x0 = (a_prev ** 0.5 * cur_b) / b_t                 # :322
        c_xt = cur_a ** 0.5 * b_prev / b_t                   # :323
        sig = self.variance(t) ** 0.5 if t > 0 else torch.tensor(0.0)
        return [float(x) for x in (a_t ** 0.5, b_t ** 0.5, c_x0, c_xt, sig)]

    def step(self, model_output, t, sample, noise=None, generator=None):  # :254-349
        t = int(t)
        prev_t = self._prev_t(t)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        b_t = 1 - a_t
        b_prev = 1 - a_prev
        cur_a = a_t / a_prev
        cur_b = 1 - cur_a
        if self.prediction_type == "epsilon":                # :299-300
            x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
        elif self.prediction_type == "sample":
            x0 = model_output
        elif self.prediction_type == "v_prediction":         # :303-304
            x0 = (a_t ** 0.5) * sample - (b_t ** 0.5) * model_output
        else:
            raise ValueError("prediction_type %s" % self.prediction_type)
        if self.clip_sample:                                 # :312-315
            x0 = x0.clamp(-self.clip_sample_range, self.clip_sample_range)
        c_x0 = (a_prev ** 0.5 * cur_b) / b_t
        c_xt = cur_a ** 0.5 * b_prev / b_t
        prev = c_x0 * x0 + c_xt * sample                     # :327
        var = 0
        if t > 0:                                            # :331-344
            if noise is None:
                noise = torch.randn(model_output.shape, generator=generator, dtype=model_output.dtype)
            var = (self.variance(t) ** 0.5) * noise
        return prev + var


class DDIMOracle:
    """fork schedulers/scheduling_ddim.py:120-360 (eta / deterministic rule)."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02,
                 beta_schedule="linear", clip_sample=True, set_alpha_to_one=True, steps_offset=0,
                 prediction_type="epsilon", clip_sample_range=1.0, eta=0.0, **_ignored):
        self.eta = eta          # default for step(): the pipeline-level argument of the reference (models.py never passes one)
        if beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                                        dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(beta_schedule)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, 0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_train_timesteps = num_train_timesteps
        self.steps_offset = steps_offset
        self.clip_sample = clip_sample
        self.clip_sample_range = clip_sample_range
        self.prediction_type = prediction_type
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))
        self.order = 1

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, n: int):                         # scheduling_ddim.py:214-236
        if n > self.num_train_timesteps:
            raise ValueError("num_inference_steps %d > num_train_timesteps %d" % (n, self.num_train_timesteps))
        self.num_inference_steps = n
        ratio = self.num_train_timesteps // n
        ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts) + self.steps_offset

    def _variance(self, t, prev_t):                          # :184-193
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        return ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)

    def step(self, model_output, t, sample, eta=None, noise=None, generator=None):  # :238-360
        t = int(t)
        eta = self.eta if eta is None else eta
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        if self.prediction_type == "epsilon":
            x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
            eps = model_output
        elif self.prediction_type == "sample":
            x0 = model_output
            eps = (sample - a_t ** 0.5 * x0) / b_t ** 0.5
        elif self.prediction_type == "v_prediction":
            x0 = (a_t ** 0.5) * sample - (b_t ** 0.5) * model_output
            eps = (a_t ** 0.5) * model_output + (b_t ** 0.5) * sample
        else:
            raise ValueError("prediction_type %s" % self.prediction_type)
        if self.clip_sample:
            x0 = x0.clamp(-self.clip_sample_range, self.clip_sample_range)
        var = self._variance(t, prev_t)
        std = eta * var ** 0.5
        direction = (1 - a_prev - std ** 2) ** 0.5 * eps
        prev = a_prev ** 0.5 * x0 + direction
        if eta > 0:
            if noise is None:
                noise = torch.randn(model_output.shape, generator=generator, dtype=model_output.dtype)
            prev = prev + std * noise
        return prev


# ----------------------------------------------------------------------------------------------
# UNet2DConditionModel (fork models/unet_2d_condition.py:520-707)
# ----------------------------------------------------------------------------------------------

def timestep_embedding(timesteps: torch.Tensor, dim: int, flip_sin_to_cos: bool, freq_shift: float,
                       max_period: int = 10000) -> torch.Tensor:
    """fork models/embeddings.py:22-62."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32)
    exponent = exponent / (half - freq_shift)
    emb = torch.exp(exponent)
    emb = timesteps[:, None].float() * emb[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


def _lin(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _conv(sd: SD, p: str, x, stride=1, padding=1):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def _gn(sd: SD, p: str, x, groups, eps):
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps)


def resnet_block_2d(sd: SD, p: str, x, temb, groups=32, eps=1e-5, groups_out=None):
    """fork models/resnet.py:549-597 (time_embedding_norm='default', no up/down, scale 1)."""
    h = F.silu(_gn(sd, p + ".norm1", x, groups, eps))
    h = _conv(sd, p + ".conv1", h)
    if temb is not None and (p + ".time_emb_proj.weight") in sd:
        h = h + _lin(sd, p + ".time_emb_proj", F.silu(temb))[:, :, None, None]
    h = F.silu(_gn(sd, p + ".norm2", h, groups_out or groups, eps))
    h = _conv(sd, p + ".conv2", h)
    if (p + ".conv_shortcut.weight") in sd:                  # resnet.py:541-547: 1x1 iff C_in != C_out
        x = _conv(sd, p + ".conv_shortcut", x, padding=0)
    return x + h


def attention(sd: SD, p: str, x, heads: int, context=None, bias=None):
    """fork models/attention_processor.py:302-337 (AttnProcessor) == :495-540 (SDPA) in fp32.

    `bias` is the additive mask [B, 1, S_kv] ((1-m)*-10000, unet_2d_condition.py:575-579), broadcast
    over heads and queries (prepare_attention_mask, attention_processor.py:263-299)."""
    ctx = x if context is None else context
    q = F.linear(x, sd[p + ".to_q.weight"])
    k = F.linear(ctx, sd[p + ".to_k.weight"])
    v = F.linear(ctx, sd[p + ".to_v.weight"])
    B, S, C = q.shape
    d = C // heads
    q = q.view(B, S, heads, d).transpose(1, 2)
    k = k.view(B, -1, heads, d).transpose(1, 2)
    v = v.view(B, -1, heads, d).transpose(1, 2)
    scores = torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5)
    if bias is not None:
        scores = scores + bias[:, None, :, :]
    probs = scores.softmax(dim=-1)
    o = torch.matmul(probs, v).transpose(1, 2).reshape(B, S, C)
    return _lin(sd, p + ".to_out.0", o)


def basic_transformer_block(sd: SD, p: str, x, heads, enc, enc_bias):
    """fork models/attention.py:276-335 (+ GEGLU :412-433, exact erf GELU)."""
    C = x.shape[-1]
    h = F.layer_norm(x, (C,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5)
    x = attention(sd, p + ".attn1", h, heads) + x
    h = F.layer_norm(x, (C,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)
    x = attention(sd, p + ".attn2", h, heads, enc, enc_bias) + x
    h = F.layer_norm(x, (C,), sd[p + ".norm3.weight"], sd[p + ".norm3.bias"], 1e-5)
    g = _lin(sd, p + ".ff.net.0.proj", h)
    val, gate = g.chunk(2, dim=-1)
    h = val * F.gelu(gate)
    return _lin(sd, p + ".ff.net.2", h) + x


def transformer_2d(sd: SD, p: str, x, heads, enc, enc_bias, groups=32):
    """fork models/transformer_2d.py:214-321 (use_linear_projection=True, GroupNorm eps 1e-6)."""
    B, C, H, W = x.shape
    res = x
    h = _gn(sd, p + ".norm", x, groups, 1e-6)
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    h = _lin(sd, p + ".proj_in", h)
    h = basic_transformer_block(sd, p + ".transformer_blocks.0", h, heads, enc, enc_bias)
    h = _lin(sd, p + ".proj_out", h)
    h = h.reshape(B, H, W, C).permute(0, 3, 1, 2).contiguous()
    return h + res


def _mask_bias(mask, dtype):
    """bool mask [B, L] (True = keep) -> additive bias [B, 1, L] (unet_2d_condition.py:575-579; _music.py:603-623)"""
    if mask is None:
        return None
    return ((1 - mask.to(dtype)) * -10000.0).unsqueeze(1)


def unet_forward(sd: SD, cfg: dict, sample, timestep, encoder_hidden_states,
                 encoder_attention_mask=None, prefix: str = "", beat_features=None, chord_features=None,
                 beat_attention_mask=None, chord_attention_mask=None) -> torch.Tensor:
    """fork models/unet_2d_condition.py:520-707 for the Tango configuration, and -- when `beat_features` / `chord_features` are
    given -- Mustango's UNet2DConditionModelMusic.forward (models/unet_2d_condition_music.py:536-757): every cross-attention
    site runs THREE Transformer2DModels in sequence, `attentions` on the text, `attentions2` on the beat embeddings and
    `attentions3` on the chord embeddings, each with its own mask bias (unet_2d_blocks.py:1199-1260 CrossAttnDownBlock2DMusic,
    :715-757 UNetMidBlock2DCrossAttnMusic, :2372-2436 CrossAttnUpBlock2DMusic).

    sample [B2,8,256,16] NCHW; timestep python int / 0-d tensor; encoder_hidden_states [B2,L,d];
    encoder_attention_mask bool [B2,L] (True = keep); beat / chord features [B2,Lb|Lc,d] with bool masks."""
    music = beat_features is not None
    cfg = normalize_unet_config(cfg)
    chans = cfg["block_out_channels"]
    heads = cfg["attention_head_dim"]
    groups, eps = cfg["norm_num_groups"], cfg["norm_eps"]
    P = prefix
    conds = [("attentions", encoder_hidden_states, _mask_bias(encoder_attention_mask, sample.dtype))]
    if music:
        conds.append(("attentions2", beat_features, _mask_bias(beat_attention_mask, sample.dtype)))
        conds.append(("attentions3", chord_features, _mask_bias(chord_attention_mask, sample.dtype)))

    def xattn(h, site, j, nheads):
        for name, feats, bias in conds:
            h = transformer_2d(sd, f"{P}{site}.{name}.{j}", h, nheads, feats, bias, groups)
        return h

    ts = torch.as_tensor(timestep).reshape(-1).expand(sample.shape[0])         # :586-600
    t_emb = timestep_embedding(ts, chans[0], cfg["flip_sin_to_cos"], cfg["freq_shift"])
    emb = _lin(sd, P + "time_embedding.linear_1", t_emb)     # embeddings.py:200-212
    emb = _lin(sd, P + "time_embedding.linear_2", F.silu(emb))

    h = _conv(sd, P + "conv_in", sample)
    skips = [h]
    for i, btype in enumerate(cfg["down_block_types"]):      # unet_2d_blocks.py:935-1074,1273-1349
        for j in range(cfg["layers_per_block"]):
            h = resnet_block_2d(sd, f"{P}down_blocks.{i}.resnets.{j}", h, emb, groups, eps)
            if btype in ("CrossAttnDownBlock2D", "CrossAttnDownBlock2DMusic"):
                h = xattn(h, f"down_blocks.{i}", j, heads[i])
            skips.append(h)
        if i != len(chans) - 1:                              # resnet.py:164-208, stride 2 pad 1
            h = _conv(sd, f"{P}down_blocks.{i}.downsamplers.0.conv", h, stride=2, padding=1)
            skips.append(h)
    # mid: unet_2d_blocks.py:492-598
    h = resnet_block_2d(sd, P + "mid_block.resnets.0", h, emb, groups, eps)
    h = xattn(h, "mid_block", 0, heads[-1])
    h = resnet_block_2d(sd, P + "mid_block.resnets.1", h, emb, groups, eps)
    rheads = list(reversed(heads))
    for i, btype in enumerate(cfg["up_block_types"]):        # unet_2d_blocks.py:2112-2248,2442-2513
        for j in range(cfg["layers_per_block"] + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = resnet_block_2d(sd, f"{P}up_blocks.{i}.resnets.{j}", h, emb, groups, eps)
            if btype in ("CrossAttnUpBlock2D", "CrossAttnUpBlock2DMusic"):
                h = xattn(h, f"up_blocks.{i}", j, rheads[i])
        if i != len(chans) - 1:                              # resnet.py:95-161 nearest x2 + conv
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(sd, f"{P}up_blocks.{i}.upsamplers.0.conv", h)
    h = F.silu(_gn(sd, P + "conv_norm_out", h, groups, eps))  # :697-700
    return _conv(sd, P + "conv_out", h)


# ----------------------------------------------------------------------------------------------
# AudioDiffusion.inference loop (models.py:210-257)
# ----------------------------------------------------------------------------------------------

def denoise_loop(sd: SD, cfg: dict, scheduler, prompt_embeds, boolean_prompt_mask, latents,
                 num_steps: int, guidance_scale: float, noises: Optional[Sequence[torch.Tensor]] = None,
                 prefix: str = "", callback=None, music: Optional[dict] = None) -> torch.Tensor:
    """models.py:224-249 (and mustango/models.py:540-598 when `music` = dict(beat_features, chord_features, beat_attention_mask,
    chord_attention_mask), all ordered [uncond; cond] like the text embeddings).  `prompt_embeds` is [2B,L,d] ordered [uncond; cond] when guidance > 1
    (models.py:301), `latents` [B,8,256,16] is draw #1 (models.py:259-264) already scaled by
    init_noise_sigma, `noises[i]` is the randn drawn inside scheduler.step at loop index i (only
    consumed when t > 0)."""
    cfg_on = guidance_scale > 1.0
    scheduler.set_timesteps(num_steps)
    for i, t in enumerate(scheduler.timesteps):
        inp = torch.cat([latents] * 2) if cfg_on else latents
        inp = scheduler.scale_model_input(inp, t)
        out = unet_forward(sd, cfg, inp, t, prompt_embeds, boolean_prompt_mask, prefix, **(music or {}))
        if cfg_on:
            u, c = out.chunk(2)
            out = u + guidance_scale * (c - u)
        n = None if noises is None else noises[i]
        latents = scheduler.step(out, t, latents, noise=n)
        if callback is not None:
            callback(i, int(t), latents)
    return latents


# ----------------------------------------------------------------------------------------------
# mel-VAE decoder (audioldm/variational_autoencoder/autoencoder.py:60-64,116-124; modules.py)
# ----------------------------------------------------------------------------------------------

def _swish(x):                                               # modules.py:33-35
    return x * torch.sigmoid(x)


def _vae_resblock(sd: SD, p: str, x):
    """modules.py:155-175 (temb None, nin_shortcut 1x1 when C_in != C_out), GN eps 1e-6 (:38-41)."""
    h = _conv(sd, p + ".conv1", _swish(_gn(sd, p + ".norm1", x, 32, 1e-6)))
    h = _conv(sd, p + ".conv2", _swish(_gn(sd, p + ".norm2", h, 32, 1e-6)))
    if (p + ".nin_shortcut.weight") in sd:
        x = _conv(sd, p + ".nin_shortcut", x, padding=0)
    return x + h


def _vae_attn(sd: SD, p: str, x):
    """modules.py:204-230: single head, d = C, scale C^-0.5."""
    h = _gn(sd, p + ".norm", x, 32, 1e-6)
    q = _conv(sd, p + ".q", h, padding=0)
    k = _conv(sd, p + ".k", h, padding=0)
    v = _conv(sd, p + ".v", h, padding=0)
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, hh * ww)
    w_ = torch.bmm(q, k) * (int(c) ** (-0.5))
    w_ = F.softmax(w_, dim=2)
    v = v.reshape(b, c, hh * ww)
    h = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(sd, p + ".proj_out", h, padding=0)


def vae_decode_first_stage(sd: SD, cfg: dict, z: torch.Tensor, prefix: str = "") -> torch.Tensor:
    """autoencoder.py:116-124 decode_first_stage -> :60-64 decode -> modules.py:650-683 Decoder.forward.
    z [B,8,256,16] -> mel [B,1,1024,64]."""
    P = prefix
    nres = len(cfg["ch_mult"])
    z = 1.0 / cfg["scale_factor"] * z                        # autoencoder.py:121
    z = _conv(sd, P + "post_quant_conv", z, padding=0)       # autoencoder.py:61
    D = P + "decoder."
    h = _conv(sd, D + "conv_in", z)
    h = _vae_resblock(sd, D + "mid.block_1", h)
    h = _vae_attn(sd, D + "mid.attn_1", h)
    h = _vae_resblock(sd, D + "mid.block_2", h)
    for lvl in reversed(range(nres)):
        for b in range(cfg["num_res_blocks"] + 1):
            h = _vae_resblock(sd, f"{D}up.{lvl}.block.{b}", h)
        if lvl != 0:                                         # modules.py:53-57
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(sd, f"{D}up.{lvl}.upsample.conv", h)
    h = _swish(_gn(sd, D + "norm_out", h, 32, 1e-6))
    return _conv(sd, D + "conv_out", h)


# ----------------------------------------------------------------------------------------------
# mel-VAE encoder (autoencoder.py:52-58,112-113,126-135; modules.py:419-543; distributions.py:24-41)
# ----------------------------------------------------------------------------------------------

def vae_encode_moments(sd: SD, cfg: dict, mel: torch.Tensor, prefix: str = "") -> torch.Tensor:
    """autoencoder.py:52-58 encode (subband 1: freq_split_subband is the identity) -> modules.py:519-543 Encoder.forward ->
    quant_conv.  mel [B,1,1024,64] -> moments [B, 2*embed_dim, 256, 16] (= [mean | logvar] along dim 1)."""
    P = prefix
    E = P + "encoder."
    nres = len(cfg["ch_mult"])
    h = _conv(sd, E + "conv_in", mel)
    for lvl in range(nres):
        for b in range(cfg["num_res_blocks"]):
            h = _vae_resblock(sd, f"{E}down.{lvl}.block.{b}", h)     # attn_resolutions is empty in the Tango VAE config
        if lvl != nres - 1:                                           # modules.py:87-91: asymmetric (0,1,0,1) pad, k3 s2 p0
            h = _conv(sd, f"{E}down.{lvl}.downsample.conv", F.pad(h, (0, 1, 0, 1), mode="constant", value=0.0), stride=2, padding=0)
    h = _vae_resblock(sd, E + "mid.block_1", h)
    h = _vae_attn(sd, E + "mid.attn_1", h)
    h = _vae_resblock(sd, E + "mid.block_2", h)
    h = _conv(sd, E + "conv_out", _swish(_gn(sd, E + "norm_out", h, 32, 1e-6)))
    return _conv(sd, P + "quant_conv", h, padding=0)


def vae_posterior(moments: torch.Tensor):
    """distributions.py:24-31: (mean, std) with logvar clamped to [-30, 20]."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    return mean, torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))


def vae_get_first_stage_encoding(moments: torch.Tensor, cfg: dict, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """autoencoder.py:126-135 on a DiagonalGaussianDistribution: scale_factor * (mean + std * noise); `noise` is the randn the
    reference draws inside `sample()` (distributions.py:37-41), None = the posterior mode."""
    mean, std = vae_posterior(moments)
    z = mean if noise is None else mean + std * noise
    return cfg["scale_factor"] * z


# ----------------------------------------------------------------------------------------------
# HiFi-GAN generator (audioldm/hifigan/models.py:96-165) + int16 cast (hifigan/utilities.py:76-86)
# ----------------------------------------------------------------------------------------------

def hifigan_generator(sd: SD, cfg: dict, mel: torch.Tensor, prefix: str = "vocoder.") -> torch.Tensor:
    """mel [B,64,T] -> wav [B,1,T*prod(rates)+...] (weight-norm already removed, utilities.py:71)."""
    P = prefix
    ks, ds = cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"]
    x = F.conv1d(mel, sd[P + "conv_pre.weight"], sd[P + "conv_pre.bias"], padding=3)
    for i, (u, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        x = F.leaky_relu(x, 0.1)
        x = F.conv_transpose1d(x, sd[f"{P}ups.{i}.weight"], sd[f"{P}ups.{i}.bias"], stride=u,
                               padding=(k - u) // 2)
        xs = None
        for j in range(len(ks)):
            rp = f"{P}resblocks.{i * len(ks) + j}"
            r = x
            for m, d in enumerate(ds[j]):                    # models.py:96-103
                xt = F.leaky_relu(r, 0.1)
                xt = F.conv1d(xt, sd[f"{rp}.convs1.{m}.weight"], sd[f"{rp}.convs1.{m}.bias"],
                              dilation=d, padding=(ks[j] * d - d) // 2)
                xt = F.leaky_relu(xt, 0.1)
                xt = F.conv1d(xt, sd[f"{rp}.convs2.{m}.weight"], sd[f"{rp}.convs2.{m}.bias"],
                              padding=(ks[j] - 1) // 2)
                r = xt + r
            xs = r if xs is None else xs + r
        x = xs / len(ks)
    x = F.leaky_relu(x)                                      # models.py:161 default slope 0.01
    x = F.conv1d(x, sd[P + "conv_post.weight"], sd[P + "conv_post.bias"], padding=3)
    return torch.tanh(x)


def wav_to_int16(wav: torch.Tensor) -> np.ndarray:
    """hifigan/utilities.py:81: (wav.cpu().numpy() * 32768).astype('int16') -- C truncation."""
    with np.errstate(invalid="ignore"):
        return (wav.cpu().numpy() * 32768).astype("int16")


def decode_to_waveform(sd: SD, hcfg: dict, mel: torch.Tensor, prefix: str = "vocoder.") -> np.ndarray:
    """autoencoder.py:66-69: mel [B,1,T,64] -> squeeze(1).permute(0,2,1) -> vocoder -> int16 [B, n]."""
    m = mel.squeeze(1).permute(0, 2, 1)
    wav = hifigan_generator(sd, hcfg, m, prefix).squeeze(1)
    return wav_to_int16(wav)


# ----------------------------------------------------------------------------------------------
# end-to-end (tango.py:43-49)
# ----------------------------------------------------------------------------------------------

def generate(unet_sd: SD, unet_cfg: dict, vae_sd: SD, vae_cfg: dict, hifi_cfg: dict, sched_cfg: dict,
             prompt_embeds, boolean_prompt_mask, latents, num_steps, guidance_scale, noises=None):
    sched = DDPMOracle(**sched_cfg)
    lat = denoise_loop(unet_sd, unet_cfg, sched, prompt_embeds, boolean_prompt_mask, latents, num_steps,
                       guidance_scale, noises)
    mel = vae_decode_first_stage(vae_sd, vae_cfg, lat)
    wav = decode_to_waveform(vae_sd, hifi_cfg, mel)
    return lat, mel, wav
